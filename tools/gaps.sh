#!/bin/bash
# usage (GPU box): tools/gaps.sh <tag> -- kernel trace of tools/step_only.py (async teacher), idle gaps of the last 6 steps
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/trace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_$tag --output-format csv -- python3 tools/step_only.py 8 > gpurun_out/trace_$tag.log 2>&1
grep '^{' gpurun_out/trace_$tag.log
f=$(ls gpurun_out/trace_$tag/*/*_kernel_trace.csv | head -1)
python3 tools/scratch/trace_gaps.py $f 6
rm -f $f
