cd $GRAFT_REPO_ROOT
for g in 1 6; do
COSA_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_check.py --out gpurun_out/ddp_dbg_$g --steps 3 --crop 64 --batch 2 --n-iter 1000000 --defer-groups $g --grid-policy 0 --log-hooks > gpurun_out/ddp_dbg_$g.log 2>&1
python - <<PY
import torch
r=torch.load("gpurun_out/ddp_dbg_$g/rank0.pt")
t0=None
for e in r["events"]:
    if e[0]=="step": print("step", e[1]); t0=None; continue
    if t0 is None: t0=e[2]
    print("  ", e[0], e[1], round((e[2]-t0)*1e3,2), "ms")
PY
done
