#!/bin/bash
# usage: pmc_collect.sh <tag> <python tool>      (on the GPU box; counters in their own runs, kernel-trace only)
set -e
tag=$1; tool=$2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  d=gpurun_out/pmc_$tag/pass_$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $d --output-format csv -- python3 $tool > $d.log 2>&1
done
