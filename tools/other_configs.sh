#!/bin/bash
# The other configurations and the sweeps of profiles/r0N_other_configs.md on the current build (run inside
# gpurun).  One line per run: flags, Mpixel/s, ms/step, kernel times, stuffed bytes per pixel.
run() {
  timeout -k 10 400 python -u bench.py "$@" > gpurun_out/oc.log 2>&1 || { echo "FAILED: $*"; tail -5 gpurun_out/oc.log; exit 1; }
  python - "$*" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/oc.log") if x.startswith("{")][-1]
d = json.loads(l); k = d["kernel_ms_per_step"]
cb = d.get("cpu_baseline") or {}
print("%-58s %9.0f Mpx/s %8.3f ms  enc %.3f cnt %.3f wr %.3f img %.3f scan %.3f  %.4f B/px  frac %.4f pipe %.4f  mismatch %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], k["k_encode_chunks"], k["k_stuff_count"], k["k_stuff_write"],
    k.get("k_stuff_image", 0.0), k["k_scan"], d["out_bytes_per_pixel"], d["roofline"]["frac"],
    d["roofline"]["pipeline"]["frac"], cb.get("mismatching_frames")))
PY
}
run --frames 1 --steps 20 --cpu-frames 1
run --frames 1 --steps 20 --content noise --cpu-frames 1
run --frames 256 --width 3840 --height 2160 --cpu-frames 4
run --frames 8 --width 16384 --height 2048 --cpu-frames 1
run --content noise --no-cpu
for p in 0.85 1.0 1.3 1.7 2.2 3.0 4.0 5.0; do run --frames 256 --persistence $p --cpu-frames 4; done
for q in 4 16 50; do run --frames 256 --quant $q --cpu-frames 8; done
