#!/bin/bash
# Rehearsal of the multi-rank bench modes on a one-GPU box: every rank uses GPU 0, the exchange goes over gloo
# (JPEGENC_BENCH_REHEARSAL=1).  Checks the code path and the result, not the speed.  usage: rehearse_ranks.sh [ranks]
n=${1:-4}
export JPEGENC_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $n "$@" > gpurun_out/rehearse.log 2> gpurun_out/rehearse.err || { tail -5 gpurun_out/rehearse.err; echo "FAILED: $*"; exit 1; }
  grep '^{' gpurun_out/rehearse.log | cut -c1-330
}
run --steps 2 --warmup 1 --frames 16 --cpu-frames 4
run --steps 2 --warmup 1 --scaling strong --total-frames 64 --cpu-frames 4
run --steps 2 --warmup 1 --mode stripes16k --side 4096
run --steps 2 --warmup 1 --mode mono16k --side 4096
