#!/bin/bash
# round-6 PMC refresh: the kernels that run NOW -- persistent GEMM at fc1 (plain bf16 and fp16c4), attention forward, the
# plane-pair PAR kernels, the round-3 lattice kernels.  Counters in their own passes (FETCH_SIZE | WRITE_SIZE | SQ sets), kernel trace only.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
which=${1:-all}
if [ "$which" = all ] || [ "$which" = gemm ]; then
bash tools/pmc_collect.sh gemm4 tools/gemm_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_gemm4 gemm_bf16_v6_kernel 679870464 "fc1+GELU M=87904 N=3072 K=768 (bf16 operands)" gpurun_out/r06_gemm_v6_pmc.json > /dev/null || exit 1
bash tools/pmc_collect.sh gemmc4 tools/gemm_c4_one.py || exit 1
# algorithmic bytes: X rows (hi 2K + blocks K) + scales, W rows likewise, c4 rows out (2N + N) + scales
python3 tools/pmc_summary.py gpurun_out/pmc_gemmc4 gemm_bf16_v6_kernel $((87904*768*3 + 87904*768/16 + 3072*768*3 + 87904*3072*3 + 87904*3072/16)) "fc1+GELU M=87904 N=3072 K=768 (fp16c4 operands, c4 rows out)" gpurun_out/r06_gemm_c4_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_gemm4 gpurun_out/pmc_gemmc4
fi
if [ "$which" = all ] || [ "$which" = gemmc8 ]; then
# the default teacher's dominant launch since round 5: fp16c8 fc1 + GELU (c8 rows in and out: hi 2K + lo8 K + hi8 K per row)
bash tools/pmc_collect.sh gemmc8 tools/gemm_c8_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_gemmc8 gemm_bf16_v6_kernel $((87904*768*4 + 3072*768*4 + 87904*3072*4)) "fc1+GELU M=87904 N=3072 K=768 (fp16c8 operands, c8 rows out)" gpurun_out/r06_gemm_c8_pmc.json > /dev/null || exit 1
bash tools/pmc_collect.sh gemmc8f2 "tools/gemm_c8_one.py fc2" || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_gemmc8f2 gemm_bf16_v6_kernel $((87904*3072*4 + 768*3072*4 + 87904*768*8)) "fc2 + residual M=87904 N=768 K=3072 (fp16c8 operands, fp32 stream in place)" gpurun_out/r06_gemm_c8_fc2_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_gemmc8 gpurun_out/pmc_gemmc8f2
fi
if [ "$which" = all ] || [ "$which" = gemmx3 ]; then
# the default teacher's dominant launch since round 6: fp16x3 fc1 + GELU (split rows in and out: hi 2K + lo 2K per row) and fc2 + residual
bash tools/pmc_collect.sh gemmx3 tools/gemm_x3_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_gemmx3 gemm_bf16_v6_kernel $((87904*768*4 + 3072*768*4 + 87904*3072*4)) "fc1+GELU M=87904 N=3072 K=768 (fp16x3 operands: hi + lo fp16 halves, split rows out)" gpurun_out/r06_gemm_x3_pmc.json > /dev/null || exit 1
bash tools/pmc_collect.sh gemmx3f2 "tools/gemm_x3_one.py fc2" || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_gemmx3f2 gemm_bf16_v6_kernel $((87904*3072*4 + 768*3072*4 + 87904*768*8)) "fc2 + residual M=87904 N=768 K=3072 (fp16x3 operands, fp32 stream in place)" gpurun_out/r06_gemm_x3_fc2_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_gemmx3 gpurun_out/pmc_gemmx3f2
fi
if [ "$which" = all ] || [ "$which" = attnx3 ]; then
bash tools/pmc_collect.sh attnx3 tools/bench_attn_x3.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_attnx3 attn_fwd_x3_kernel $((32*1765*12*64*2*2*4)) "B=32 N=1765 H=12 (teacher scale 1.5), fp16x3 operands (q, k, v, out as hi + lo halves)" gpurun_out/r06_attn_x3_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_attnx3
fi
if [ "$which" = all ] || [ "$which" = attn ]; then
bash tools/pmc_collect.sh attn4 tools/attn_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_attn4 attn_fwd2_kernel 433766400 "B=32 N=1765 H=12 (teacher scale 1.5), fp16 operands, no-grad variant (flag bit 10), 4 waves per workgroup" gpurun_out/r06_attn_fwd_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_attn4
fi
if [ "$which" = all ] || [ "$which" = gemmfc2 ]; then
bash tools/pmc_collect.sh gemmc4f2 "tools/gemm_c4_one.py fc2" || exit 1
# algorithmic bytes: X rows (hi 2K + blocks K) + scales, W rows, fp32 residual in and out
python3 tools/pmc_summary.py gpurun_out/pmc_gemmc4f2 gemm_bf16_v6_kernel $((87904*3072*3 + 87904*3072/16 + 768*3072*3 + 87904*768*8)) "fc2 + residual M=87904 N=768 K=3072 (fp16c4 operands, fp32 stream in place)" gpurun_out/r06_gemm_c4_fc2_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_gemmc4f2
fi
if [ "$which" = all ] || [ "$which" = wgrad ]; then
bash tools/pmc_collect.sh wgb4 tools/wgrad_batched_one.py || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_wgb4 gemm_wgrad_batched 3708616704 "12 blocks x (qkv, proj, fc1, fc2) at M=12560: 1296 tiles of 256 x 256, one launch" gpurun_out/r06_wgrad_batched_pmc.json > /dev/null || exit 1
rm -rf gpurun_out/pmc_wgb4
fi
if [ "$which" = all ] || [ "$which" = lattice ]; then
bash tools/pmc_kernels.sh lattice4 tools/bench_bilateral.py "" > gpurun_out/pmc_lattice4.txt 2>&1 || exit 1
python3 tools/lattice_pmc_summary.py gpurun_out/pmc_lattice4.json gpurun_out/r06_lattice_pmc.json 13 > gpurun_out/r06_lattice_pmc.txt || exit 1
rm -rf gpurun_out/pmc_lattice4
fi
if [ "$which" = all ] || [ "$which" = par ]; then
rm -rf gpurun_out/pmc_par4; mkdir -p gpurun_out/pmc_par4
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=gpurun_out/pmc_par4/pass_$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $d --output-format csv -- python3 tools/par_one.py > $d.log 2>&1 || echo "pass failed: $set"
done
python3 - <<PY
import csv, glob, json
tot, cnt = {}, {}
for f in glob.glob("gpurun_out/pmc_par4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "par_" not in k: continue
        k = k.replace("cosa::(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]
        key = (k, r["Counter_Name"])
        tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"]); cnt[key] = cnt.get(key, 0) + 1
info = json.loads([l for l in open("gpurun_out/pmc_par4/pass_FETCH_SIZE.log") if l.startswith("{")][-1])
passes = info["passes"]
kern = {}
for (k, c), v in tot.items():
    kern.setdefault(k, {})[c] = {"per_dispatch": round(v / cnt[(k, c)], 2), "dispatches_per_pass": cnt[(k, c)] / passes}
hbm = sum((tot.get((k, "FETCH_SIZE"), 0.0) * 2 + tot.get((k, "WRITE_SIZE"), 0.0)) * 1024 for k in kern) / passes
out = {"_comment": "rocprofv3 --pmc passes of tools/par_one.py (cam2mask_multi with PAR(T=10, dilations 1,2,4,8,12,24), main + aux CAM sets, b=16, 448^2; "
                   "one pass = the 4 PAR calls per image of the metric).  FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
       "passes": passes, "mean_K": info["mean_K"], "hbm_bytes_per_pass": int(hbm), "algorithmic_bytes_per_pass": int(info["algorithmic_bytes_per_pass"]),
       "kernels": kern}
for k, c in kern.items():
    f, w = c.get("FETCH_SIZE", {}).get("per_dispatch", 0) * 2048, c.get("WRITE_SIZE", {}).get("per_dispatch", 0) * 1024
    c["fetch_MB_per_dispatch"], c["write_MB_per_dispatch"] = round(f / 1e6, 2), round(w / 1e6, 2)
    h, m = c.get("TCC_HIT_sum", {}).get("per_dispatch", 0), c.get("TCC_MISS_sum", {}).get("per_dispatch", 0)
    c["l2_hit"] = round(h / (h + m), 3) if h + m else None
json.dump(out, open("gpurun_out/r06_par_pmc.json", "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("kernels", "_comment")})
for k, c in kern.items(): print(k, c["fetch_MB_per_dispatch"], c["write_MB_per_dispatch"], c["l2_hit"], c.get("FETCH_SIZE", {}).get("dispatches_per_pass"))
PY
rm -rf gpurun_out/pmc_par4
fi
python3 -c "
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_*_pmc.json')):
    d=json.load(open(f)); print(f, d.get('hbm_bytes_per_launch', d.get('hbm_bytes_per_pass', d.get('hbm_bytes_per_forward_backward'))), d.get('algorithmic_bytes_per_launch', d.get('algorithmic_bytes_per_pass')), d.get('mfma_busy_frac_of_simd_cycles'))
"
