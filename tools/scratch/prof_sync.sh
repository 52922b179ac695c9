#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
export COSA_TEACHER_SYNC=1
rm -rf gpurun_out/prof_sync
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sync --output-format csv -- python3 tools/step_only.py 10 > gpurun_out/prof_sync.log 2>&1 || exit 1
grep -h '^{' gpurun_out/prof_sync.log
python3 tools/summarize_prof.py gpurun_out/prof_sync gpurun_out/r03_syncstep_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 tools/step_only.py 10 (COSA_TEACHER_SYNC=1: teacher and student serialised, kernel durations not inflated by sharing)" > /dev/null
rm -f gpurun_out/prof_sync/*/*_kernel_trace.csv
