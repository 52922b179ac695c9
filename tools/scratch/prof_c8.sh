#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
export COSA_TEACHER_SYNC=1
rm -rf gpurun_out/prof_c8
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c8 --output-format csv -- python3 tools/step_only.py 10 fp16c8-9 > gpurun_out/prof_c8.log 2>&1 || exit 1
grep -h '^{' gpurun_out/prof_c8.log
python3 tools/summarize_prof.py gpurun_out/prof_c8 gpurun_out/r03_c8step_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 tools/step_only.py 10 fp16c8-9 (COSA_TEACHER_SYNC=1: teacher serialised; final round-3 tree)" > /dev/null
rm -f gpurun_out/prof_c8/*/*_kernel_trace.csv
