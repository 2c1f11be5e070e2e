#!/bin/bash
# Several builds of the library on one box over several contents: one line per build and content, two rounds.
# usage (inside gpurun): AB_BUILDS="base new" bash tools/ab_multi.sh "<bench flags 1>" "<bench flags 2>" ...
# (builds are ab/<name>.so; default "base new").  Leaves ab/new.so installed.
builds=${AB_BUILDS:-base new}
for flags in "$@"; do
  for v in $builds $builds; do
    cp ab/$v.so jpeg-encoder_amd/libjpegenc.so
    timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu $flags > gpurun_out/abm.log 2>&1 || { tail -5 gpurun_out/abm.log; exit 1; }
    python - "$v" "$flags" <<'PY'
import json, sys
l = [x for x in open("gpurun_out/abm.log") if x.startswith("{")][-1]
d = json.loads(l); k = d["kernel_ms_per_step"]
print("%-5s %-44s %9.0f Mpx/s %8.3f ms  enc %.3f cnt %.3f wr %.3f img %.3f  %.3f B/px" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], k["k_encode_chunks"], k["k_stuff_count"], k["k_stuff_write"],
    k.get("k_stuff_image", 0.0), d["out_bytes_per_pixel"]))
PY
  done
done
cp ab/new.so jpeg-encoder_amd/libjpegenc.so
