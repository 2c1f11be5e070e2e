#!/bin/bash
# Small calls (1 to 64 frames of 1080p) without event recording, as one kernel launch (default) and as separate
# kernels (--split); run inside gpurun.  us per call and Mpixel/s.
for f in 1 2 4 8 16; do
  for v in "" "--split"; do
    timeout -k 10 200 python -u bench.py --frames $f --steps 200 --warmup 20 --no-cpu --no-profile $v $LAT_ARGS > gpurun_out/lat.log 2>&1 || { tail -5 gpurun_out/lat.log; exit 1; }
    python - "$f" "$v" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/lat.log") if x.startswith("{")][-1]
d = json.loads(l)
print("frames %3s %-8s %8.1f us/call %9.0f Mpx/s" % (sys.argv[1], sys.argv[2] or "single", 1000 * d["ms_per_step"], d["value"]))
PY
  done
done
