#!/bin/bash
# same-box A/B of the attention kernels: current library vs tools/scratch/libcosa_head.so (built from HEAD)
for rep in 1 2; do
for lib in cur head; do
  echo "== $lib"
  python - <<PY
import os, sys
sys.path.insert(0, '.')
from cosa_amd import _C
if "$lib" == "head":
    _C.LIB_PATH = os.path.abspath("tools/scratch/libcosa_head.so")
sys.argv = ["x"]
src = open("tools/scratch/attn_augm.py").read()
exec(compile(src, "attn_augm", "exec"))
exec(compile(open("tools/bench_attn.py").read(), "bench_attn", "exec"))
PY
done; done 2>&1 | grep -v amdgpu | grep "== \|float16   flags=0x400\|bfloat16  flags=0x000 B=16\|bwd" | cut -c1-70
