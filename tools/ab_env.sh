#!/bin/bash
# usage (GPU box): tools/ab_env.sh "ENV=VAL ..." ["ENV=VAL ..." ...] -- bench.py (30 steps) of this tree under each environment, interleaved twice
B="--steps 30 --warmup 6 --no-cpu-baseline --no-parity-grade"
for rep in 1 2; do
  for e in "$@"; do
    env $e python bench.py $B > gpurun_out/abe.json || exit 1
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/abe.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])" "[$e]"
  done
done
