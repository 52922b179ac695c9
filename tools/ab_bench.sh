#!/bin/bash
# usage (GPU box): tools/ab_bench.sh -- bench.py (30 steps) in the ./_r01 checkout and in this tree, same box.
# ./_r01 is another commit of this repository with its library built (it travels to the GPU box with the snapshot; it is git-ignored):
#   git worktree add _r01 <commit> && (cd _r01 && python -m cosa_amd.build)        ... and afterwards: git worktree remove --force _r01
B="--steps 30 --warmup 6 --no-cpu-baseline"
p() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])" $1 "$2"; }
(cd _r01 && python bench.py $B > ../gpurun_out/ab0.json) && p gpurun_out/ab0.json r01 || exit 1
python bench.py $B --no-parity-grade > gpurun_out/ab1.json && p gpurun_out/ab1.json r02 || exit 1
