#!/bin/bash
# Scaling curve on one 8 x MI355X node with the flags the driver uses (bench.py contract): N = 1, 2, 4, 8 back to back, one rank per GPU
# over RCCL.  Prints one JSON line per N; efficiency is computed by the reader from the per-N values.
#   tools/run_scale.sh [steps] [warmup]
set -e
STEPS=${1:-50}; WARMUP=${2:-10}; PORT=${PORT:-29511}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps $STEPS --warmup $WARMUP
done
