#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by more than most changes are worth): put the
# builds in ab/base.so and ab/new.so, run `bash tools/ab_bench.sh` inside gpurun; prints ms/step and the
# encode kernel's ms for three alternating rounds (AB_ARGS adds bench options, e.g. AB_ARGS="--quant 16").  Leaves ab/new.so installed.
for round in 1 2 3; do
  for v in base new; do
    cp ab/$v.so jpeg-encoder_amd/libjpegenc.so
    timeout -k 10 200 python -u bench.py --frames 256 --steps 20 --warmup 3 --no-cpu $AB_ARGS > gpurun_out/ab_$v.log 2>&1 || { tail -5 gpurun_out/ab_$v.log; exit 1; }
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_$v.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$v", $round, d["ms_per_step"], d["kernel_ms_per_step"]["k_encode_chunks"])
PY
  done
done
