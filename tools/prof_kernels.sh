#!/bin/bash
# usage (GPU box): tools/prof_kernels.sh <tag> <tool.py> [ENV=VALUE ...] -- rocprofv3 kernel stats of a tool, top kernels printed
tag=$1; tool=$2; shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python3 $tool > gpurun_out/prof_$tag.log 2>&1
grep '^{' gpurun_out/prof_$tag.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("$tag", r["Name"][:70].replace("cosa::(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
