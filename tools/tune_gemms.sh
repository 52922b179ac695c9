#!/bin/bash
# Re-create cosa_amd/tuning/tunableop_gfx950.csv: let PyTorch's TunableOp time the hipBLASLt / rocBLAS solutions for every library
# GEMM shape of the benchmark step (student forward / input gradients / narrow heads), then keep the result file.
#   bash tools/tune_gemms.sh            (on a GPU box, from the repo root; ~1 minute)
set -e
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_VERBOSE=0
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 10 --warmup 8 > gpurun_out/tune_bench.json
cp gpurun_out/tunableop_results0.csv cosa_amd/tuning/tunableop_gfx950.csv
echo "wrote cosa_amd/tuning/tunableop_gfx950.csv"
