#!/bin/bash
# Small launches (1 to 64 frames) on two builds, ab/base.so and ab/new.so (run inside gpurun): ms per step.
# Leaves ab/new.so installed.
for f in 1 4 16 32 64; do
  for v in base new; do
    cp ab/$v.so jpeg-encoder_amd/libjpegenc.so
    timeout -k 10 200 python -u bench.py --frames $f --steps 50 --warmup 5 --no-cpu --no-profile > gpurun_out/lat_$v.log 2>&1 || { tail -5 gpurun_out/lat_$v.log; exit 1; }
    python - <<PY
import json
l=[x for x in open("gpurun_out/lat_$v.log") if x.startswith("{")][-1]
d=json.loads(l)
print("frames", $f, "$v", "us/step", round(1000*d["ms_per_step"],1), "Mpx/s", d["value"])
PY
  done
done
