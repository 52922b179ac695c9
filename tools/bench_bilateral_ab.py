"""A/B of the dense-energy regulariser (tools/bench_bilateral.py) between two builds of the library, each in a fresh child process, interleaved.
usage: python tools/bench_bilateral_ab.py cosa_amd/lib/libcosa_hip_old.so cosa_amd/lib/libcosa_hip.so [rounds=3]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
code = ("import sys, runpy; sys.path.insert(0, %r); import cosa_amd._C as C; C.LIB_PATH = sys.argv[1]; sys.argv = ['bench_bilateral.py']; "
        "runpy.run_path(%r, run_name='__main__')") % (root, os.path.join(root, "tools", "bench_bilateral.py"))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", code, os.path.abspath(l)], capture_output=True, text=True, cwd=root)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(out.stderr[-2000:]); sys.exit(1)
        res[l].append(json.loads(line[-1])["bilateral_fwd_bwd_ms_per_img"])
for l in libs:
    print(os.path.basename(l), "ms/img:", res[l], "min", min(res[l]))
