#!/bin/bash
# As pmc_pass.sh with the number of frames per launch as second argument (512 and more take k_stuff_image).
# usage: pmc_pass_n.sh <tag> <frames> <counters...>
tag=$1; frames=$2; shift; shift
export TMPDIR=/tmp
# the counter list goes into the log first: a profiler that dies names no counters itself
echo "counters: $*" > gpurun_out/pmc_${tag}.log
timeout -s KILL 280 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag} -- python3 bench.py --frames $frames --steps 2 --warmup 1 --no-cpu --no-profile $PMC_BENCH_ARGS >> gpurun_out/pmc_${tag}.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_${tag}/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "jpegenc" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k[:60])
        for c, v in agg[k].items(): print("   %-28s %.4g per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
