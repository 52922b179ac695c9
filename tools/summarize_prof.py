"""Condense a rocprofv3 --kernel-trace --stats CSV into a small text summary for profiles/."""
import csv, sys, glob, os
src = sys.argv[1]; dst = sys.argv[2]; note = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary\n# source: {f}\n# {note}\n# total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches\n")
    o.write("total_ms,pct,calls,avg_us,min_us,max_us,kernel\n")
    for r in rows[:60]:
        o.write(f"{float(r['TotalDurationNs'])/1e6:.3f},{float(r['Percentage']):.2f},{r['Calls']},{float(r['AverageNs'])/1e3:.1f},{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},\"{r['Name'][:140]}\"\n")
print(open(dst).read()[:6000])
