"""largest idle gaps between consecutive kernels in the last K steps of a rocprofv3 kernel trace (single stream view)"""
import csv, sys, re
src, K = sys.argv[1], int(sys.argv[2])
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(src))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
lo, hi = marks[-K - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
gaps = []
cur_end = sel[0][1]
for i in range(1, len(sel)):
    s, e, n = sel[i]
    if s > cur_end:
        gaps.append((s - cur_end, sel[i - 1][2][:60], n[:60]))
    cur_end = max(cur_end, e)
tot = sum(g[0] for g in gaps)
print(f"total idle {tot/1e6/K:.3f} ms/step in {len(gaps)/K:.0f} gaps/step")
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
for g in sorted(gaps, reverse=True)[:14]:
    print(f"{g[0]/1e3:8.1f} us  after {short(g[1])}  before {short(g[2])}")
