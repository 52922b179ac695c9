"""Two-rank rehearsal of the REAL trainer (CoSATrainer with DistributedDataParallel) for tests/test_distributed_gpu.py.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/ddp_check.py --out DIR ...
    python tools/ddp_check.py --single --out DIR ...          (one process, the two ranks' batches concatenated)

Backend: $COSA_DIST_BACKEND (default nccl == RCCL; the one-GPU boxes rehearse with gloo, both ranks on card 0).  Every rank runs `--steps`
training steps on its own synthetic shard (seed 100 + rank) and saves its student and teacher parameters."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--crop", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--n-iter", type=int, default=1, help="iteration number passed to the step (<= warmup_iters: classification losses only)")
    ap.add_argument("--single", action="store_true")
    ap.add_argument("--ranks", type=int, default=2, help="--single: how many ranks' shards to concatenate")
    ap.add_argument("--defer-groups", type=int, default=0, help="batched weight-gradient launches per backward pass (0: the trainer's choice)")
    ap.add_argument("--grid-policy", type=int, default=-1, help="persistent-GEMM grid policy (-1: the trainer's choice)")
    ap.add_argument("--log-hooks", action="store_true", help="record when the DeferredWgrad nodes run and when DDP's bucket hooks fire")
    ap.add_argument("--teacher-graph", action="store_true", help="the teacher pass as a captured hipGraph on a side stream, as in bench.py (the "
                    "capture happens before DistributedDataParallel is constructed: CoSATrainer.prepare_ddp); the run FAILS if it was not captured")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and wrap the student in DDP even in a world of one "
                    "(one rank over RCCL: the watchdog thread, the comm stream and the bucket all-reduces are the real ones)")
    opt = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")) if not opt.single else 1
    rank = int(os.environ.get("RANK", "0")) if not opt.single else 0
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
    torch.cuda.set_device(dev)
    use_dist = world > 1 or (opt.force_dist and not opt.single)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("COSA_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    shards = [synthetic_batch(opt.batch, opt.crop, 20, dev, seed=100 + r) for r in (range(opt.ranks) if opt.single else [rank])]
    wimg, simg, lab = (torch.cat([s[i] for s in shards]) for i in range(3))
    box = torch.cat([s[3] for s in shards])
    args = default_args("VOC12", crop_size=opt.crop, batch_size=wimg.shape[0], teacher_graph=opt.teacher_graph, lr=1e-3)
    tr = CoSATrainer(args, dev, ddp=use_dist, seed=0)
    if use_dist:
        tr.prepare_ddp(wimg, lab)          # teacher capture first, then the DDP wrap
        assert isinstance(tr.model_ON, torch.nn.parallel.DistributedDataParallel)
    if opt.teacher_graph and use_dist:
        assert tr._graph is not None and tr.graph_error is None, f"teacher hipGraph was not captured under backend {dist.get_backend()}: {tr.graph_error}"
    from cosa_amd import _C, nn_ops
    if opt.defer_groups > 0:
        tr.student.encoder.defer_groups = opt.defer_groups
    if opt.grid_policy >= 0:
        _C.lib().cosa_gemm_set_grid_policy(opt.grid_policy)
        _C.lib().cosa_gemm_set_grid_policy_f16(opt.grid_policy)
    events = []
    if opt.log_hooks and use_dist:
        import time
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        nn_ops.event_log = events

        def hook(state, bucket):          # DDP calls this when a gradient bucket is complete: the point where its all-reduce starts
            events.append(("bucket", bucket.index(), time.perf_counter()))
            return default_hooks.allreduce_hook(state, bucket)
        tr.model_ON.register_comm_hook(None, hook)
    for st in range(opt.steps):
        events.append(("step", st, 0.0))
        logs = tr.step(wimg, simg, lab, box, n_iter=opt.n_iter)
    torch.cuda.synchronize()
    os.makedirs(opt.out, exist_ok=True)
    state = {"student": {k: v.detach().cpu() for k, v in tr.student.named_parameters()},
             "teacher": {k: v.detach().cpu() for k, v in tr.model_AN.named_parameters()},
             "loss": float(logs["overall_loss"]), "world": dist.get_world_size() if use_dist else 1, "events": events,
             "defer_groups": tr.student.encoder._n_defer_groups(), "graph_captured": tr._graph is not None,
             "backend": dist.get_backend() if use_dist else None, "teacher_async": tr.teacher_async}
    torch.save(state, os.path.join(opt.out, "single.pt" if opt.single else f"rank{rank}.pt"))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
