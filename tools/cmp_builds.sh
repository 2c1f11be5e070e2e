#!/bin/bash
# Several builds of the library on ONE box: put them in ab/<name>.so, run `bash tools/cmp_builds.sh <name>...`
# inside gpurun; prints ms per step, kernel ms per step and Mpixel/s of each (CMP_ARGS adds bench options,
# default "--frames 768").  Leaves the last build installed.
for v in "$@"; do
  cp ab/$v.so jpeg-encoder_amd/libjpegenc.so
  timeout -k 10 200 python -u bench.py ${CMP_ARGS:---frames 768} --steps 10 --warmup 3 --no-cpu > gpurun_out/cmp_$v.log 2>&1 || { echo "$v failed: stopping"; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/cmp_$v.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$v", d["ms_per_step"], d.get("kernel_ms_per_step"), d["value"], d.get("out_bytes_per_pixel"))
PY
done
