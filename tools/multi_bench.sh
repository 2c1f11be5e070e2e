#!/bin/bash
# Short bench of every library build in ab/c_*.so and ab/new.so on one box (AB_ARGS adds bench options): for
# bisecting a slowdown over several commits.  Leaves ab/new.so installed.
for so in ab/c_*.so ab/new.so; do
  cp $so jpeg-encoder_amd/libjpegenc.so
  timeout -k 10 200 python -u bench.py --frames 256 --steps 10 --warmup 3 --no-cpu $AB_ARGS > gpurun_out/multi.log 2>&1 || { tail -5 gpurun_out/multi.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/multi.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$so", d["ms_per_step"], d["kernel_ms_per_step"]["k_encode_chunks"])
PY
done
