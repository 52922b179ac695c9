#!/bin/bash
# usage (GPU box): tools/ab_bench.sh -- bench.py (30 steps) in the _r01 worktree and in this tree under a few switches, same box
B="--steps 30 --warmup 6 --no-cpu-baseline"
p() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])" $1 "$2"; }
(cd _r01 && python bench.py $B > ../gpurun_out/ab0.json) && p gpurun_out/ab0.json r01 || exit 1
python bench.py $B --no-parity-grade > gpurun_out/ab1.json && p gpurun_out/ab1.json r02 || exit 1
