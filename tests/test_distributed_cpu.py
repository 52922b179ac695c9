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
