#!/bin/bash
# rocprofv3 kernel stats of the dense-energy prepare (tools/scratch/lat_build.py).  The COSA_LAT_ABL values quoted in profiles/r03_results.md
# (1 = no hash probing, 2 = no offset / weight stores, 4 = ids not taken from the per-image counter) belonged to a temporary build of
# lattice_build_kernel; the shipped library ignores the variable (run with the loop reduced to a single pass).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
for a in 0; do
  export COSA_LAT_ABL=$a
  rm -rf gpurun_out/prof_lat$a
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lat$a --output-format csv -- python3 tools/scratch/lat_build.py > gpurun_out/prof_lat$a.log 2>&1 || { tail -3 gpurun_out/prof_lat$a.log; exit 1; }
  grep "^abl" gpurun_out/prof_lat$a.log
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_lat$a/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("  ", r["Name"][:60].replace("cosa::(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
