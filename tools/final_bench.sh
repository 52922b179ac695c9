#!/bin/bash
# usage (GPU box): tools/final_bench.sh -- the round's bench lines (default run + variants) and the rocprofv3 kernel stats of the default command
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err || { tail -5 gpurun_out/r02_bench_final.err; exit 1; }
tail -c 600 gpurun_out/r02_bench_final.json; echo
python bench.py --usepar --no-cpu-baseline --no-parity-grade > gpurun_out/r02_bench_usepar.json 2>/dev/null || exit 1
python bench.py --teacher-precision fp16 --no-cpu-baseline --no-parity-grade > gpurun_out/r02_bench_fp16teacher.json 2>/dev/null || exit 1
python bench.py --dataset COCO --no-cpu-baseline --no-parity-grade > gpurun_out/r02_bench_coco448.json 2>/dev/null || exit 1
python bench.py --dataset COCO --crop 640 --no-cpu-baseline --no-parity-grade > gpurun_out/r02_bench_coco640.json 2>/dev/null || exit 1
rm -rf gpurun_out/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 || exit 1
rm -f gpurun_out/prof_bench/*/*_kernel_trace.csv
for f in usepar fp16teacher coco448 coco640; do python3 -c "import json,sys; d=json.loads(open('gpurun_out/r02_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
