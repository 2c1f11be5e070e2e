#!/bin/bash
# Encode-kernel time and overflow fractions over content detail with small, large and automatically chosen LDS
# regions (run inside gpurun).  usage: region_sweep.sh [P ...]   (default 3.5 3.7 4.0 4.3 5.0 7.0)
Ps=${@:-3.5 3.7 4.0 4.3 5.0 7.0}
for P in $Ps; do for R in small large auto; do
timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --no-cpu --persistence $P --regions $R > gpurun_out/sw.log 2>&1 || { tail -5 gpurun_out/sw.log; exit 1; }
python - $P $R <<'PY'
import json, sys
l = [x for x in open("gpurun_out/sw.log") if x.startswith("{")][-1]
d = json.loads(l); k = d["kernel_ms_per_step"]
print("P %s %-5s %7.0f Mpx/s enc %.3f over %s %.3f B/px" % (sys.argv[1], sys.argv[2], d["value"], k["k_encode_chunks"], d.get("chunks_over_region"), d["out_bytes_per_pixel"]))
PY
done; done
