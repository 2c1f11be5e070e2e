#!/bin/bash
# Quick GPU iteration on the box: core parity subset, then a short bench.  Usage (inside gpurun):
#   bash tools/gpu_quick.sh [tag]
tag=${1:-quick}
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread \
  -k "config1 or ragged or natural_and_noise or extreme or quirk or batch or bands or tiny or grid" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
timeout -k 10 300 python -u bench.py --frames 256 --steps 10 --warmup 2 --no-cpu > gpurun_out/${tag}_bench.log 2>&1 || { tail -20 gpurun_out/${tag}_bench.log; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/${tag}_bench.log") if x.startswith("{")][-1]
d=json.loads(l)
print("${tag}", "Mpx/s", d["value"], "ms/step", d["ms_per_step"], "kernels", d.get("kernel_ms_per_step"), "frac", d["roofline"]["frac"])
PY
