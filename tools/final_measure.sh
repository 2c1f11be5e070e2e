#!/bin/bash
# Full GPU suite, default bench line, kernel trace and the two traffic counter passes of one build (run inside
# gpurun); the outputs under gpurun_out/ are what profiles/r0N_* and pmc_traffic.json are copied from.
export TMPDIR=/tmp
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -20 gpurun_out/final_bench.err; exit 1; }
cut -c1-400 gpurun_out/final_bench.json
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/final_trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/final_trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh fetch FETCH_SIZE | grep -A1 jpegenc
bash tools/pmc_pass.sh write WRITE_SIZE | grep -A1 jpegenc
