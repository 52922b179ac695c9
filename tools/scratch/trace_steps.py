"""Per-step kernel table from a rocprofv3 --kernel-trace CSV: only the last K training steps (the timed region of bench.py).
Step boundaries = launches of the fused AdamW+EMA kernel (exactly one per step).
usage: trace_steps.py <kernel_trace.csv> K <out.csv> [note]"""
import csv, sys, collections, re
src, K, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(src))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
assert len(marks) > K, (len(marks), K)
lo, hi = marks[-K - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
span = (sel[-1][1] - sel[0][0]) / 1e6 / K
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:110]
    agg[n][0] += e - s; agg[n][1] += 1
busy = sum(v[0] for v in agg.values()) / 1e6 / K
with open(dst, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace, last {K} steps only; {note}\n# step span {span:.2f} ms, kernel busy {busy:.2f} ms/step, {len(sel)/K:.0f} launches/step\n")
    o.write("ms_per_step,pct,launches_per_step,avg_us,kernel\n")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        o.write(f"{t/1e6/K:.3f},{100*t/1e6/K/busy:.1f},{c/K:.1f},{t/c/1e3:.1f},\"{n}\"\n")
print(open(dst).read()[:7000])
