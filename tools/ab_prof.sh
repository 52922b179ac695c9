#!/bin/bash
# usage (GPU box): tools/ab_prof.sh -- kernel stats of tools/step_only.py (teacher serialised) in this tree and in the _r01 worktree, same box
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
export COSA_TEACHER_SYNC=1
cp tools/step_only.py _r01/tools/step_only.py
rm -rf gpurun_out/ab_r02 gpurun_out/ab_r01
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_r02 --output-format csv -- python3 tools/step_only.py 10 > gpurun_out/ab_r02.log 2>&1 || exit 1
cd _r01 && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ../gpurun_out/ab_r01 --output-format csv -- python3 tools/step_only.py 10 > ../gpurun_out/ab_r01.log 2>&1 || exit 1
cd ..
grep -h '^{' gpurun_out/ab_r02.log gpurun_out/ab_r01.log
rm -f gpurun_out/ab_r0*/*/*_kernel_trace.csv
