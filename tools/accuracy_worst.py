"""worst line per teacher mode and crop of an accuracy record (profiles/rNN_accuracy_teacher.txt or gpurun_out/...): max normalised-CAM rel. err,
max own-scale err (records from round 5 on), min label agreement / mask mIoU, the draw it comes from.   usage: python tools/accuracy_worst.py [file]"""
import collections, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_accuracy_teacher.txt"
w = collections.defaultdict(lambda: dict(rel=0.0, own=None, iou=1.0, agree=1.0, n=0, where="", seeds=set(), over=set()))
for ln in open(f):
    m = re.match(r"teacher (\S+)\s+S=(\d+) b=(\d+) seed=(\d+)\s+(\S+)\s*: .*rel err (\S+)\s+label agreement (\S+)\s+mask mIoU (\S+)(?:\s+own-scale err (\S+))?", ln)
    if m:
        e = w[(m.group(1), int(m.group(2)))]
        r, ag, iou = float(m.group(6)), float(m.group(7)), float(m.group(8))
        if r > e["rel"]:
            e["rel"], e["where"] = r, f"seed {m.group(4)} b={m.group(3)} {m.group(5)}"
        if r > 1e-3:
            e["over"].add((m.group(4), m.group(3)))
        if m.group(9):
            e["own"] = max(e["own"] or 0.0, float(m.group(9)))
        e["iou"], e["agree"], e["n"] = min(e["iou"], iou), min(e["agree"], ag), e["n"] + 1
        e["seeds"].add((m.group(4), m.group(3)))
for (mode, S), e in sorted(w.items()):
    if e["own"] is not None:
        ok = e["own"] <= 1e-3 and e["iou"] >= 0.999 and e["agree"] >= 0.999
        print(f"{mode:13s} S={S} lines {e['n']:3d} draws {len(e['seeds']):2d}  own-scale err {e['own']:.3e} ({1e-3 / e['own']:.2f}x)  min agreement {e['agree']:.5f}  min mIoU {e['iou']:.5f}  "
              f"{'ok   ' if ok else 'FAILS'}  | normalised planes: worst {e['rel']:.3e} ({e['where']}), over 1e-3 on {len(e['over'])} draws")
    else:
        ok = e["rel"] <= 1e-3 and e["iou"] >= 0.999
        print(f"{mode:13s} S={S} lines {e['n']:3d} draws {len(e['seeds']):2d}  worst rel err {e['rel']:.3e} ({e['where']}; margin {1e-3 / e['rel']:.2f}x)  min mIoU {e['iou']:.5f}  {'ok' if ok else 'FAILS'}")
