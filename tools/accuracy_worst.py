"""per teacher mode and crop of an accuracy record (profiles/rNN_accuracy_teacher.txt or gpurun_out/...): the pre-registered criterion of
tests/test_precision_gpu.py as bench.conformance evaluates it -- planes / exempt / failed, the worst literal and own-scale figures, label
agreement, pooled and per-draw mask mIoU.   usage: python tools/accuracy_worst.py [file]"""
import collections
import importlib.util
import os
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
f = sys.argv[1] if len(sys.argv) > 1 else bench.newest_profile("accuracy_teacher.txt")
rows, h = bench.parse_accuracy_record(f)
print(f"# {f}: {len(rows)} lines, source hash {h} (tree: {bench.teacher_csrc_hash()})")
groups = collections.defaultdict(list)
for r in rows:
    groups[(r["mode"], r["S"])].append(r)
for (mode, S), rs in sorted(groups.items()):
    draws = {(r["seed"], r["b"]) for r in rs}
    if rs[0]["planes"] is None:
        print(f"{mode:13s} S={S} draws {len(draws):3d}  (record without per-plane bookkeeping) worst literal {max(r['rel'] for r in rs):.3e}")
        continue
    pooled = collections.defaultdict(dict)
    for r in rs:
        for c, v in r["conf"].items():
            pooled[r["set"]][c] = [x + y for x, y in zip(pooled[r["set"]].get(c, [0, 0, 0]), v)]
    pm = min(bench.pooled_miou(v)[0] for v in pooled.values())
    fail, ex, planes = sum(r["fail"] for r in rs), sum(r["exempt"] for r in rs), sum(r["planes"] for r in rs)
    fd = {(r["seed"], r["b"]) for r in rs if r["fail"] or r["agree"] < bench.AGREE_BAR}
    ok = fail == 0 and min(r["agree"] for r in rs) >= bench.AGREE_BAR and pm >= bench.MIOU_BAR
    print(f"{mode:13s} S={S} draws {len(draws):3d} planes {planes:4d} exempt {ex:2d} FAILED {fail:2d} (draws {len(fd)})  literal worst {max(r['rel'] for r in rs):.3e}  "
          f"own-scale {max(r['own'] for r in rs):.3e}  min agreement {min(r['agree'] for r in rs):.5f}  pooled mIoU {pm:.5f}  per-draw min mIoU {min(r['iou'] for r in rs):.5f}  "
          f"{'ok   ' if ok else 'FAILS'}")
    for r in rs:
        if r["fail"] or r["exempt"]:
            print(f"      seed {r['seed']} b={r['b']} {r['set']}: {r['notes']}")
