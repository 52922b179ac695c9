#!/bin/bash
# usage (GPU box): tools/pmc_kernels.sh <tag> <tool.py> <kernel-name substring> [ENV=VALUE ...]
# rocprofv3 PMC passes (counters in their own runs, --kernel-trace only) of a tool; sums per counter over the kernels whose name contains the
# substring, divided by the number of `iterations` the tool reports (JSON key "iters", default 13 = 3 warm-up + 10 timed).
tag=$1; tool=$2; sub=$3; shift; shift; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
rm -rf gpurun_out/pmc_$tag; mkdir -p gpurun_out/pmc_$tag
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=gpurun_out/pmc_$tag/pass_$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $d --output-format csv -- python3 $tool > $d.log 2>&1 || echo "pass failed: $set"
done
python3 - <<PY
import csv, glob, json
tot, calls = {}, {}
for f in glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$sub" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].replace("cosa::(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0][:40]
        tot[(k, r["Counter_Name"])] = tot.get((k, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] = calls.get((k, r["Counter_Name"]), 0) + 1
out = {}
for (k, c), v in sorted(tot.items()):
    out.setdefault(k, {})[c] = {"sum": round(v, 1), "dispatches": calls[(k, c)], "per_dispatch": round(v / calls[(k, c)], 2)}
json.dump(out, open("gpurun_out/pmc_$tag.json", "w"), indent=1)
for k in out:
    print("$tag", k, {c: out[k][c]["per_dispatch"] for c in out[k]})
PY
