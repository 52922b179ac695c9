#!/bin/bash
# same-box A/B of the attention kernels: the current library against tools/scratch/libcosa_<tag>.so (a library linked with another version of
# attn_kernels.hip: compile that version twice (bf16 / -DCOSA_OP_F16=1) and link it with the other objects of cosa_amd/lib/obj).
# usage (GPU box): tools/scratch/ab_attn.sh <tag> [<tag> ...]
for rep in 1 2; do
for lib in cur "$@"; do
  echo "== $lib"
  python - <<PY
import os, sys
sys.path.insert(0, '.')
from cosa_amd import _C
if "$lib" != "cur":
    _C.LIB_PATH = os.path.abspath("tools/scratch/libcosa_$lib.so")
sys.argv = ["x"]
exec(compile(open("tools/scratch/attn_augm.py").read(), "attn_augm", "exec"))
exec(compile(open("tools/bench_attn.py").read(), "bench_attn", "exec"))
PY
done; done 2>&1 | grep -v amdgpu | grep "== \|float16   flags=0x400\|bfloat16  flags=0x000 B=16\|bfloat16  flags=0x000 B=32 N=1765\|bwd" | cut -c1-70
