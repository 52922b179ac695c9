#!/bin/bash
# usage (GPU box): tools/par_prof.sh <tag> [ENV=VALUE ...]  -- rocprofv3 kernel stats of tools/bench_par.py, top kernels printed
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/prof_par_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_par_$tag --output-format csv -- python3 tools/bench_par.py > gpurun_out/prof_par_$tag.log 2>&1
grep '^{' gpurun_out/prof_par_$tag.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_par_$tag/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("$tag", r["Name"][:60].replace("cosa::(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
