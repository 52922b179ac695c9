"""worst line per teacher mode and crop of an accuracy record (profiles/rNN_accuracy_teacher.txt or gpurun_out/...): max normalised-CAM rel. err,
min mask mIoU, the seed it comes from, margin on the 1e-3 bar.   usage: python tools/accuracy_worst.py [file]"""
import collections, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_accuracy_teacher.txt"
w = collections.defaultdict(lambda: [0.0, 1.0, 0, "", set()])
for ln in open(f):
    m = re.match(r"teacher (\S+)\s+S=(\d+) b=\d+ seed=(\d+)\s+(\S+)\s*: .*rel err (\S+) .*mIoU (\S+)", ln)
    if m:
        k = (m.group(1), int(m.group(2)))
        r, iou = float(m.group(5)), float(m.group(6))
        if r > w[k][0]:
            w[k][0], w[k][3] = r, f"seed {m.group(3)} {m.group(4)}"
        w[k][1] = min(w[k][1], iou)
        w[k][2] += 1
        w[k][4].add(m.group(3))
for (mode, S), v in sorted(w.items()):
    ok = v[0] <= 1e-3 and v[1] >= 0.999
    print(f"{mode:13s} S={S} lines {v[2]:2d} seeds {len(v[4])}  worst rel err {v[0]:.3e} ({v[3]}; margin {1e-3 / v[0]:.2f}x)  min mIoU {v[1]:.5f}  {'ok' if ok else 'FAILS'}")
