#!/bin/bash
# usage (GPU box): tools/final_bench.sh -- the round's bench lines (default run + variants) and the rocprofv3 kernel stats of the default command
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
R=${R:-r06}
python bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err || { tail -5 gpurun_out/${R}_bench_final.err; exit 1; }
tail -c 400 gpurun_out/${R}_bench_final.json; echo
python bench.py --usepar --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_usepar.json 2>/dev/null || exit 1
python bench.py --teacher-precision fp16c8-x2 --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_fp16c8x2teacher.json 2>/dev/null || exit 1
python bench.py --dataset COCO --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_coco448.json 2>/dev/null || exit 1
# configs[4]: global batch 64 over 8 ranks = 8 per rank
python bench.py --dataset COCO --crop 640 --batch 8 --no-cpu-baseline --no-secondary > gpurun_out/${R}_bench_coco640.json 2>/dev/null || exit 1
rm -rf gpurun_out/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/prof_bench.log 2>&1 || exit 1
python3 tools/summarize_prof.py gpurun_out/prof_bench gpurun_out/${R}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary (all legs of the bench command)" > /dev/null
rm -f gpurun_out/prof_bench/*/*_kernel_trace.csv
for f in usepar fp16c8x2teacher coco448 coco640; do python3 -c "import json,sys; d=json.loads(open('gpurun_out/${R}_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
