"""CPU suite: the N>1 (data-parallel) host logic with world_size=2 over gloo -- DDP wrapper config, per-rank shards,
fused-free PolyWarmupAdamW, EMA update.  The HIP operators themselves need a GPU; the collective path does not."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cosa_amd.train_step import rank_seed, wrap_ddp
        from cosa_amd.utils import torch_helper
        torch.manual_seed(0)                       # identical init on every rank, like main.py:35
        student = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
        teacher = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
        teacher.load_state_dict(student.state_dict())
        for p in teacher.parameters():
            p.requires_grad = False
        ddp = wrap_ddp(student, torch.device("cpu"))
        opt = torch_helper.PolyWarmupAdamW([{"params": list(student[0].parameters()), "lr": 6e-5},
                                            {"params": list(student[2].parameters()), "lr": 6e-4}], lr=6e-5, weight_decay=1e-2,
                                           betas=(0.9, 0.999), warmup_iter=1500, max_iter=32000, warmup_ratio=1e-6, power=0.9)
        g = torch.Generator().manual_seed(rank_seed(1234, rank))        # distinct shard per rank
        x = torch.randn(5, 8, generator=g)
        loss = ddp(x).square().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        # the all-reduced gradient equals the mean of the per-rank gradients
        local = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
        local.load_state_dict(teacher.state_dict())
        local(x).square().mean().backward()
        mine = local[0].weight.grad.clone()
        dist.all_reduce(mine)
        assert torch.allclose(student[0].weight.grad, mine / world, atol=1e-7)
        opt.step()
        torch_helper.ema_update(list(teacher.parameters()), list(student.parameters()), 0.9994)
        flat = torch.cat([p.detach().flatten() for p in list(student.parameters()) + list(teacher.parameters())])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered)        # ranks stay bit-identical without a broadcast
        assert opt.param_groups[1]["lr"] == pytest.approx(6e-4 * 1e-6, rel=1e-9)   # warm-up step 0
        ret[rank] = float(x.sum())
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == 2 and ret[0] != ret[1]                            # the two ranks really saw different shards


class _CollectingLinear(torch.autograd.Function):
    """stand-in for nn_ops.LinearShadowFn on the CPU: same contract towards the collector (backward hands (w, dY, X) over and returns no
    weight / bias gradient of its own)"""

    @staticmethod
    def forward(ctx, x, w, b):
        from cosa_amd import nn_ops
        ctx.save_for_backward(x, w)
        c = nn_ops._wgrad_collector
        ctx.collect = (c, w) if (c is not None and id(w) in c.keys) else None
        if ctx.collect is not None:
            c.mark_used(w)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        assert ctx.collect is not None
        ctx.collect[0].add(ctx.collect[1], dy.reshape(-1, dy.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1]).contiguous())
        return dy @ w, None, None


class _DeferredMLP(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.inp = torch.nn.Linear(8, 16)
        self.l1, self.l2, self.l3 = torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4)

    def forward(self, x, defer, groups=1):
        """groups: the collected linears in `groups` runs, each with its own DeferredWgrad node at the run's input (the trainer's form
        under data parallelism: VisionTransformer.defer_groups)"""
        from cosa_amd import nn_ops
        h = torch.tanh(self.inp(x))
        if not defer:
            return self.l3(torch.tanh(self.l2(torch.tanh(self.l1(h)))))
        runs = {1: [[self.l1, self.l2, self.l3]], 2: [[self.l1], [self.l2, self.l3]], 3: [[self.l1], [self.l2], [self.l3]]}[groups]
        for run in runs:
            h, c = nn_ops.defer_wgrads(h, run)
            with nn_ops.collecting(c):
                for m in run:
                    h = _CollectingLinear.apply(h, m.weight, m.bias)
                    if m is not self.l3:
                        h = torch.tanh(h)
        return h


def _worker_deferred(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cosa_amd import nn_ops
        from cosa_amd.train_step import rank_seed, wrap_ddp
        # the batched kernel needs a GPU: its torch equivalent (fp32) stands in, everything around it is the product's code
        nn_ops.gemm_wgrad_batched = lambda pairs: [(dy.float().t() @ x.float(), dy.float().sum(0) if wb else None) for dy, x, wb in pairs]
        torch.manual_seed(0)
        model = _DeferredMLP()
        ref = _DeferredMLP()
        ref.load_state_dict(model.state_dict())
        ddp = wrap_ddp(model, torch.device("cpu"))
        x = torch.randn(6, 8, generator=torch.Generator().manual_seed(rank_seed(77, rank)))
        for it in range(6):                                      # DDP rebuilds its buckets after the first iteration; iterations 3-5 in the grouped forms
            model.zero_grad(set_to_none=True)
            ddp(x, True, (1, 1, 1, 2, 3, 3)[it]).square().mean().backward()
            ref.zero_grad(set_to_none=True)
            ref(x, False).square().mean().backward()
            for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                g = q.grad.clone()
                dist.all_reduce(g)
                assert p.grad is not None and torch.allclose(p.grad, g / world, atol=1e-6), (it, n)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_deferred_weight_gradients_reach_ddp_world_size_2_gloo():
    """nn_ops.DeferredWgrad under DistributedDataParallel: the gradients of the collected linears come out of ONE autograd node at the
    region's input -- or, grouped (VisionTransformer.defer_groups, the trainer's form when world > 1), out of one node per run of layers;
    DDP must still see every parameter exactly once per iteration and average it over the ranks (run on CPU over gloo, the batched kernel
    replaced by its torch equivalent)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_deferred, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == 2


def test_deferred_wgrad_node_in_the_wrong_place_raises():
    """a DeferredWgrad node whose backward runs BEFORE the backward of a linear it collects for would drop that weight gradient silently:
    it raises instead (nn_ops.DeferredWgrad)"""
    from cosa_amd import nn_ops
    old = nn_ops.gemm_wgrad_batched
    nn_ops.gemm_wgrad_batched = lambda pairs: [(dy.float().t() @ x.float(), dy.float().sum(0) if wb else None) for dy, x, wb in pairs]
    try:
        torch.manual_seed(0)
        l1, l2 = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
        x = torch.randn(4, 8, requires_grad=True)
        h = _CollectingLinear.apply(x, l1.weight, l1.bias) if False else x
        # the node sits BEHIND l1 (on l1's output) although it collects for l1 too: its backward comes first
        c = nn_ops.WgradCollector([l1.weight, l2.weight])
        with nn_ops.collecting(c):
            h = _CollectingLinear.apply(h, l1.weight, l1.bias)
            h = nn_ops.DeferredWgrad.apply(h, c, l1.weight, l1.bias, l2.weight, l2.bias)
            h = _CollectingLinear.apply(h, l2.weight, l2.bias)
        with pytest.raises(RuntimeError, match="DeferredWgrad"):
            h.square().mean().backward()
    finally:
        nn_ops.gemm_wgrad_batched = old
