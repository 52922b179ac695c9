#!/bin/bash
# round-3 PMC refresh: attention forward at N = 1765 (4-wave workgroups now) and the batched weight gradient
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
bash tools/pmc_collect.sh attn3 tools/attn_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_attn3 attn_fwd2_kernel 433766400 "B=32 N=1765 H=12 (teacher scale 1.5), 4 waves per workgroup" gpurun_out/r03_attn_fwd_pmc.json > /dev/null || exit 1
bash tools/pmc_collect.sh wgb3 tools/scratch/wgrad_batched_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_wgb3 gemm_wgrad_batched 3708616704 "12 blocks x (qkv, proj, fc1, fc2) at M=12560: 2592 tiles, one launch" gpurun_out/r03_wgrad_batched_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_attn3/*/*/*kernel_trace* gpurun_out/pmc_wgb3/*/*/*kernel_trace*
python3 -c "
import json
for f in ('r03_attn_fwd_pmc','r03_wgrad_batched_pmc'):
    d=json.load(open('gpurun_out/'+f+'.json')); print(f, d.get('hbm_bytes_per_launch'), d.get('algorithmic_bytes_per_launch'), d.get('mfma_busy_frac_of_simd_cycles'), d.get('SQ_LDS_BANK_CONFLICT'))
"
