#!/bin/bash
# Full GPU suite, default bench line, kernel trace and the counter passes of one build (run inside gpurun); the
# outputs under gpurun_out/ are what profiles/r0N_* and pmc_traffic.json are copied from.  The counter passes run
# the bench with worst-case chunk slots, so that every launch of a pass has the same size (with slots sized from
# experience the first call sends a first group of images ahead).
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -20 gpurun_out/final_bench.err; exit 1; }
cut -c1-400 gpurun_out/final_bench.json
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/final_trace.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/final_trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
export PMC_BENCH_ARGS="--slots worst --regions short"  # (the size the default bench settles on; three launches do not get there on their own)
bash tools/pmc_pass.sh fetch FETCH_SIZE > gpurun_out/pmc_fetch.txt 2>&1; grep -A1 jpegenc gpurun_out/pmc_fetch.txt
bash tools/pmc_pass.sh write WRITE_SIZE > gpurun_out/pmc_write.txt 2>&1; grep -A1 jpegenc gpurun_out/pmc_write.txt
bash tools/pmc_pass.sh valu SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU > gpurun_out/pmc_valu.txt 2>&1; cat gpurun_out/pmc_valu.txt
bash tools/pmc_pass_n.sh valu512 512 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU > gpurun_out/pmc_valu512.txt 2>&1; cat gpurun_out/pmc_valu512.txt
