"""gpurun_out/pmc_<tag>.json (tools/pmc_kernels.sh, all kernels of tools/bench_bilateral.py) -> profiles/r02_lattice_pmc.json: per-kernel HBM
traffic of the dense-energy regulariser's forward + backward (lattice_* kernels and the rocPRIM sort of the splat lists).
usage: python tools/lattice_pmc_summary.py gpurun_out/pmc_lattice.json profiles/r02_lattice_pmc.json [iterations=13]"""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 13
raw = json.load(open(src))
out = {"_comment": "rocprofv3 --pmc passes (counters in their own runs + --kernel-trace) of tools/bench_bilateral.py: get_energy_loss forward+backward, "
                   "b=16, 448^2 crops (lattice at 224^2), K=21; lattice_* kernels and the radix sort of the splat lists. FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md; per-dispatch figures are per kernel launch over the 16-image batch", "kernels": {}}
total = 0.0
for k, c in sorted(raw.items()):
    if not ("lattice_" in k or "rocprim" in k or "radix" in k or "energy_" in k or "half_denorm" in k):
        continue
    g = lambda n: c.get(n, {}).get("per_dispatch", 0.0)
    disp = max(v["dispatches"] for v in c.values())
    fetch, write = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    out["kernels"][k] = {"dispatches": disp, "fetch_MB_per_dispatch": round(fetch / 1e6, 2), "write_MB_per_dispatch": round(write / 1e6, 2),
                         "atomics_per_dispatch": round(g("TCC_ATOMIC_sum"), 2), "l2_hit": round(hit / (hit + miss), 3) if hit + miss else None,
                         "wait_frac_of_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAVE_CYCLES") else None}
    total += (fetch + write) * disp / iters
out["hbm_bytes_per_forward_backward"] = int(total)
out["hbm_MB_per_image"] = round(total / 16 / 1e6, 1)
out["compulsory_MB_per_image"] = round(4.0 * 224 * 224 * (3 + 2 * 21) / 1e6, 2)
out["iterations"] = iters
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for k, v in out["kernels"].items():
    print(k, v)
