#!/bin/bash
# N-rank rehearsal of bench.py's strong-scaling mode on a ONE-GPU box: every rank on GPU 0, gloo instead of RCCL
# (device buffers staged through the host). usage: tools/rehearse.sh <ranks> <log2n> [extra bench.py flags]
set -o pipefail
N=${1:-2}; L=${2:-20}; shift 2
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --backend gloo --device 0 --log2n $L --steps 2 --warmup 1 --no-cpu-baseline "$@"
