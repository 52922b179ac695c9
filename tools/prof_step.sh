#!/bin/bash
# rocprofv3 --kernel-trace --stats of N training steps in one teacher mode, teacher serialised (kernel durations not stretched by sharing
# the GPU); summary -> gpurun_out/<tag>_kernel_stats.csv.   usage: tools/prof_step.sh <mode> <tag> [steps=10] [extra step_only args]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
mode=${1:?mode}; tag=${2:?tag}; steps=${3:-10}; shift 3 2>/dev/null
export COSA_TEACHER_SYNC=1
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python3 tools/step_only.py $steps $mode "$@" > gpurun_out/prof_$tag.log 2>&1 || exit 1
grep -h '^{' gpurun_out/prof_$tag.log
python3 tools/summarize_prof.py gpurun_out/prof_$tag gpurun_out/${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 tools/step_only.py $steps $mode $* (COSA_TEACHER_SYNC=1: teacher serialised)" > /dev/null
rm -rf gpurun_out/prof_$tag
