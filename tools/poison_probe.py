"""does any kernel of the training step read memory it (or a predecessor) did not write?  Fill the allocator's free blocks with NaN / garbage
before the step and compare the loss and a block-0 gradient with a clean run.   usage: python tools/poison_probe.py none|nan|<value>
(round 4: identical bits for none / nan / 1000 -- profiles/r04_poison_probe.txt)"""
import sys, os, torch, numpy as np
sys.path.insert(0, '.')
from cosa_amd import _C
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda", 0)
S = 224
poison = sys.argv[1]
args = default_args("VOC12", crop_size=S, teacher_precision="fp16c8", teacher_graph=False, teacher_async=False)
tr = CoSATrainer(args, dev, seed=3)
wimg, simg, lab, box = synthetic_batch(2, S, 20, dev, seed=5)
n_iter = args.warmup_iters + 1
if poison != "none":
    val = float("nan") if poison == "nan" else float(poison)
    junk = [torch.full((256 * 1024 * 1024,), val, device=dev) for _ in range(8)]      # 8 GiB in 1-GiB blocks
    small = [torch.full((n,), val, device=dev) for n in (1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 24) for _ in range(16)]
    del junk, small
loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
tr.optimizer.zero_grad(set_to_none=True)
loss.backward()
named = dict(tr.student.named_parameters())
g0 = named["encoder.blocks.0.attn.qkv.weight"].grad.float()
print(poison, "loss", float(loss), "g0 norm", g0.norm().item(), "g0 checksum", g0.double().sum().item(), "nan grads:",
      [n for n, p in named.items() if p.grad is not None and not torch.isfinite(p.grad).all()][:5])
