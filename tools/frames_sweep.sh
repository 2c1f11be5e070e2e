#!/bin/bash
# Per-frame kernel times at several batch sizes (run inside gpurun; FRAMES="..." chooses the sizes).
for f in ${FRAMES:-32 64 128 256 384 512 768 1024 1536}; do
  timeout -k 10 200 python -u bench.py --frames $f --steps 10 --warmup 3 --no-cpu > gpurun_out/fr_$f.log 2>&1 || { tail -5 gpurun_out/fr_$f.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/fr_$f.log") if x.startswith("{")][-1]
d=json.loads(l); k=d["kernel_ms_per_step"]
print($f, "ms/step", d["ms_per_step"], "per-frame us:", {a: round(1000*b/$f,3) for a,b in k.items()})
PY
done
