#!/bin/bash
# usage (GPU box): tools/par_pmc.sh <tag> [ENV=VALUE ...] -- rocprofv3 PMC passes (counters only + kernel trace) of tools/bench_par.py
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/pmc_par_$tag; mkdir -p gpurun_out/pmc_par_$tag
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmc_par_$tag/pass_$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $d --output-format csv -- python3 tools/bench_par.py > $d.log 2>&1 || echo "pass failed: $set"
done
python3 - <<PY
import csv, glob
tot, cnt = {}, {}
for f in glob.glob("gpurun_out/pmc_par_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "par_" not in k: continue
        k = k.split("(")[0].replace("cosa::(anonymous namespace)::", "").replace("void ", "")[:40]
        key = (k, r["Counter_Name"])
        tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"]); cnt[key] = cnt.get(key, 0) + 1
for (k, c) in sorted(tot):
    print("$tag", k, c, round(tot[(k, c)] / cnt[(k, c)], 1))
PY
