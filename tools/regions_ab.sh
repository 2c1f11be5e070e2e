#!/bin/bash
# The installed library with the encode kernel's LDS regions forced to each of the given sizes in turn (run inside
# gpurun): ms per step, kernel ms per step, Mpixel/s.  usage: regions_ab.sh "<bench args>" size [size ...]
ARGS=$1; shift
for r in "$@"; do
  timeout -k 10 200 python -u bench.py $ARGS --regions $r --no-cpu > gpurun_out/rab.log 2>&1 || { tail -5 gpurun_out/rab.log; exit 1; }
  python - $r <<'PY'
import json, sys
l = [x for x in open("gpurun_out/rab.log") if x.startswith("{")][-1]
d = json.loads(l); k = d["kernel_ms_per_step"]
print("%-6s %.3f ms/step  enc %.4f  stuff %.4f  %.0f Mpx/s" % (sys.argv[1], d["ms_per_step"], k["k_encode_chunks"], k["k_stuff_image"] + k["k_stuff_count"] + k["k_stuff_write"] + k["k_scan"], d["value"]))
PY
done
