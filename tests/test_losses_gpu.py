"""GPU parity of the loss kernels and of the bf16 student's own-kernel autograd path against the ORACLE and the reference's golden vectors
(round 2's tests of these kernels compared them with this repository's own op-by-op torch path only).

  * msm_loss_kernel / cam_target_kernel            vs tests/golden/misc.npz {refine_out, camloss_out} (reference: utils/seg_helper.py:553-602)
  * seg_loss_fwd/bwd + lattice (dense energy)      vs oracle/torch_oracle.py {seg_loss, energy_loss_and_grad} on the same inputs
                                                   (the oracle is pinned to misc.npz / bilateral.npz by tests/test_oracle_golden.py)
  * the whole bf16 training step at ViT-B          vs oracle/cpu_step.py (fp32 CPU) on identical weights and inputs: five losses and the
                                                   gradients of qkv / fc1 / conv6 / classifier weights
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_hip_cam_loss_kernels_vs_reference_golden(golden):
    """cam_loss of the reference (seg_helper.py:593-602) = multilabel soft margin of relu(cam) against the bilinearly resized
    seg_refine_by_label output.  (a) msm_loss_kernel on the reference's own resized targets reproduces `camloss_out` to 1e-6 relative;
    (b) cam_target_kernel, fed the golden seg as a single full-size 'scale' (+ a zero flipped half), reproduces the resized `refine_out`;
    (c) both kernels chained reproduce `camloss_out`."""
    from cosa_amd.utils import seg_helper
    g = golden("misc")
    dev = "cuda"
    cam = torch.from_numpy(g["camloss_cam"]).to(dev)
    ref_ps = torch.from_numpy(g["refine_out"]).to(dev)
    labels = torch.from_numpy(g["refine_labels"]).to(dev)
    tgt = F.interpolate(ref_ps[:, 1:], size=cam.shape[-2:], mode="bilinear", align_corners=False).contiguous()
    loss = seg_helper.cam_loss_from_targets(cam, tgt)                       # the fused HIP value + gradient kernel (fp32 CUDA logits)
    assert float(loss) == pytest.approx(float(g["camloss_out"]), rel=1e-6)
    # gradient of the kernel against autograd of the torch expression
    cam_t = cam.clone().requires_grad_(True)
    cam_k = cam.clone().requires_grad_(True)
    F.multilabel_soft_margin_loss(F.relu(cam_t).permute(0, 2, 3, 1).reshape(-1, cam.shape[1]), tgt.permute(0, 2, 3, 1).reshape(-1, cam.shape[1])).backward()
    seg_helper.cam_loss_from_targets(cam_k, tgt).backward()
    assert torch.allclose(cam_k.grad, cam_t.grad, rtol=1e-5, atol=1e-8)
    seg = torch.from_numpy(g["refine_seg"]).to(dev)
    B, K, S, _ = seg.shape
    scales = [torch.cat([seg, torch.zeros_like(seg)], 0).contiguous()]     # full = seg + flip(0) = seg
    out = seg_helper.cam_loss_targets(scales, labels, S, tuple(cam.shape[-2:]), 0.01)
    assert torch.allclose(out, tgt, atol=2e-6, rtol=1e-5), (out - tgt).abs().max().item()
    assert float(seg_helper.cam_loss_from_targets(cam, out)) == pytest.approx(float(g["camloss_out"]), rel=1e-5)
    # the public functions on the device (torch expressions) against the same vectors
    assert torch.allclose(seg_helper.seg_refine_by_label(seg, labels, 0.01), ref_ps, rtol=1e-4, atol=2e-6)     # (torch's device softmax at T = 0.01)
    assert float(seg_helper.cam_loss(cam, ref_ps)) == pytest.approx(float(g["camloss_out"]), rel=1e-5)
    pred, mask = torch.from_numpy(g["segloss_pred"]).to(dev), torch.from_numpy(g["segloss_mask"]).to(dev)
    assert float(seg_helper.seg_loss(pred, mask)) == pytest.approx(float(g["segloss_out"]), rel=1e-5)


@pytest.mark.parametrize("K,hs,S,boxes", [(21, 12, 192, [[0, 192, 0, 192], [10, 150, 20, 190]]), (6, 4, 64, [[0, 64, 0, 64], [3, 60, 0, 50]])])
def test_hip_seg_and_energy_loss_vs_oracle(K, hs, S, boxes):
    """seg_loss x 2 (main + aux labels, blended 0.5 / 0.5, main.py:200-203) and the dense-energy regulariser (seg_helper.py:199-230,
    864-903) from the fused HIP kernels against oracle/torch_oracle.py on the same inputs: values 1e-4 / 1e-3 relative, the gradient
    with respect to the low-resolution logits 2e-3 of its maximum (the regulariser's gradient is the reference's hand-written one)."""
    from cosa_amd.utils import seg_helper
    from oracle import torch_oracle as to
    torch.manual_seed(K)
    rng = np.random.default_rng(K)
    B = len(boxes)
    seg_lr = (torch.randn(B, K, hs, hs) * 2)
    vals = [0, 1, K - 1, 255] if K < 8 else [0, 1, 5, K - 1, 255]
    mk = lambda: torch.from_numpy(rng.choice(vals, size=(B, S, S)).astype(np.float32))
    mA, mB = mk(), mk()
    simg = torch.randn(B, 3, S, S)
    box = torch.tensor(boxes, dtype=torch.int16)
    w_seg, w_reg = 0.1, 0.05
    # oracle (CPU, fp32)
    lr_o = seg_lr.clone().requires_grad_(True)
    up = F.interpolate(lr_o, size=(S, S), mode="bilinear", align_corners=False)
    l_seg = 0.5 * to.seg_loss(up, mA) + 0.5 * to.seg_loss(up, mB)
    l_reg, g_up = to.energy_loss_and_grad(simg, up.detach(), mA.to(torch.uint8).unsqueeze(1), box.numpy())
    (w_seg * l_seg).backward(retain_graph=True)
    up.backward(w_reg * g_up)
    # HIP
    lr_g = seg_lr.cuda().requires_grad_(True)
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    f_seg, f_reg = seg_helper.fused_seg_and_energy_loss(lr_g, mA.cuda(), mB.cuda(), simg.cuda(), box, layer)
    (w_seg * f_seg + w_reg * f_reg).sum().backward()
    assert float(f_seg) == pytest.approx(float(l_seg), rel=1e-4)
    assert float(f_reg) == pytest.approx(float(l_reg), rel=1e-3, abs=1e-12)
    err = (lr_g.grad.cpu() - lr_o.grad).abs().max().item()
    assert err <= 2e-3 * lr_o.grad.abs().max().item(), (err, lr_o.grad.abs().max().item())


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


# weight-gradient cosine bars of the student against the fp32 CPU oracle, per residual-stream mode (measured values: profiles/r04_student_vs_oracle_*.txt)
#   fp32 stream (the default): what is left is the rounding of the MFMA operands (bf16 activations / weights / dY, 8 significant bits)
#   bf16 stream (rounds 1-3, kept for A/B): the stream and every gradient sum rounded to 8 bits as well
#   The cosines are ONE draw of the rounding noise: a forward attention kernel that differed from the present one in the last bit of 3 of 302 592
#   outputs (round 4, a reordered row sum) re-rolled every bf16 rounding downstream and moved block 0's qkv cosine from 0.9964 to 0.9937 at
#   S = 224 (394 tokens per gradient); at the benchmark's S = 448 (1570 tokens) the same weights sit at 0.997-0.9999.  Hence 0.995 at 448 and
#   0.992 at 224; the student's attention kernel is kept bit-stable (tests/test_network_gpu.py) so the record in profiles/ stays comparable.
STUDENT_BARS = {"fp32": dict(enc=0.995, enc224=0.992, qk=0.99, dec=0.999, loss=1e-3), "bf16": dict(enc=0.98, enc224=0.98, qk=0.97, dec=0.99, loss=3e-3)}


@pytest.mark.parametrize("S,stream", [(448, "fp32"), (224, "bf16")])          # (224 / fp32: five seeds of it in test_student_gradient_cosines_over_seeds_vs_cpu_oracle)
def test_bf16_student_step_vs_cpu_oracle_vitb(S, stream):
    """The student's forward / backward through this repository's own kernels (bf16-operand GEMMs with fused epilogues -- the projections
    that close a residual branch add into the FP32 stream in their epilogue --, fused attention forward / backward, LayerNorm forward /
    backward on the fp32 stream, dilated convs, narrow heads, fused losses) at ViT-B against oracle/cpu_step.py (fp32 CPU) on identical
    weights and inputs, b = 2, all five losses live, at S = 224 and at the benchmark's S = 448.  The teacher runs in the parity-grade mode
    (fp16c8), so both sides see the same pseudo labels; what is compared is the student: the five losses within BASELINE.json's 1e-3
    relative, weight gradients by cosine (bars: STUDENT_BARS)."""
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    from oracle.cpu_step import CpuStep
    dev = torch.device("cuda", 0)
    b, C = 2, 20
    bars = STUDENT_BARS[stream]
    args = default_args("VOC12", crop_size=S, teacher_precision="fp16c8", teacher_graph=False, teacher_async=False)
    tr = CoSATrainer(args, dev, seed=3)
    tr.student.encoder.residual_stream = stream
    sd = {k: v.detach().cpu().clone() for k, v in tr.student.state_dict().items()}
    wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=5)
    n_iter = args.warmup_iters + 1
    loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    cpu = CpuStep(sd, num_classes=21, aux_layer=-4)
    closs, clogs = cpu.losses(wimg.cpu(), simg.cpu(), lab.cpu(), box.numpy(), n_iter)
    agree = (logs["mask"].cpu().numpy() == clogs["mask"].numpy()).mean()
    assert agree >= 0.999, f"label agreement {agree}"
    lines = []
    for k in ("cls_loss", "cls_aux_loss", "seg_loss", "cam_loss", "reg_loss", "overall_loss"):
        a, c = float(logs[k]), float(clogs[k])
        lines.append(f"{k}: hip {a:.6f} oracle {c:.6f}")
        assert a == pytest.approx(c, rel=bars["loss"], abs=2e-5), (k, a, c)
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    cpu.opt.zero_grad(set_to_none=True)
    closs.backward()
    named = dict(tr.student.named_parameters())
    checks = []
    for name in ("encoder.blocks.0.attn.qkv.weight", "encoder.blocks.6.attn.qkv.weight", "encoder.blocks.11.attn.qkv.weight",
                 "encoder.blocks.0.attn.proj.weight", "encoder.blocks.5.mlp.fc1.weight", "encoder.blocks.0.mlp.fc2.weight",
                 "encoder.blocks.11.mlp.fc2.weight", "encoder.patch_embed.proj.weight", "decoder.conv6.weight", "decoder.conv7.weight",
                 "decoder.conv8.weight", "classifier.weight", "aux_classifier.weight", "encoder.blocks.3.norm1.weight"):
        gg, gc = named[name].grad.float().cpu(), cpu.student.p(name).grad
        parts = [("", gg, gc)]
        if name.endswith("qkv.weight"):          # query / key / value rows separately: at near-uniform attention (random initialisation) the
            parts += [(":" + n, gg[i * 768:(i + 1) * 768], gc[i * 768:(i + 1) * 768]) for i, n in enumerate("qkv")]      # q / k gradients are tiny
        for tag, a, c in parts:
            cs, ratio = _cos(a, c), float(a.norm() / (c.norm() + 1e-30))
            lines.append(f"grad {name}{tag}: cosine {cs:.5f} norm ratio {ratio:.4f} (oracle norm {float(c.norm()):.3e})")
            checks.append((name + tag, cs, ratio))
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"r04_student_vs_oracle_{stream}_{S}.txt"), "w") as f:
        f.write(f"# student ({stream} residual stream, bf16 MFMA operands) vs oracle/cpu_step.py (fp32 CPU), ViT-B, S = {S}, b = 2, one step\n")
        f.write("\n".join(lines) + "\n")
    for name, cs, ratio in checks:
        bar = bars["qk"] if name.endswith((":q", ":k")) else ((bars["enc"] if S >= 448 else bars["enc224"]) if "encoder.blocks" in name or "patch_embed" in name
                                                              else bars["dec"])
        assert cs >= bar and 0.95 <= ratio <= 1.05, (name, cs, ratio)


STUDENT_SEEDS = (3, 7, 11, 19, 23)


def test_student_gradient_cosines_over_seeds_vs_cpu_oracle():
    """The student's weight-gradient cosines against oracle/cpu_step.py are ONE draw of the bf16 rounding noise per (weights, batch): a
    single-draw bar pins the kernels' last bits rather than their accuracy.  This is the statistic instead: five
    weight / batch seeds at S = 224 (394 tokens per gradient: the noisier crop), fp32 residual stream, the encoder weights of
    test_bf16_student_step_vs_cpu_oracle_vitb -- MEAN and MINIMUM over the seeds per weight, losses within 1e-3 on every seed.  A kernel
    change that re-rolls the noise moves single draws by a few 1e-3 and leaves these where they are; a real loss of accuracy does not."""
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    from oracle.cpu_step import CpuStep
    dev = torch.device("cuda", 0)
    S, b, C = 224, 2, 20
    names = ("encoder.blocks.0.attn.qkv.weight", "encoder.blocks.6.attn.qkv.weight", "encoder.blocks.11.attn.qkv.weight",
             "encoder.blocks.0.attn.proj.weight", "encoder.blocks.5.mlp.fc1.weight", "encoder.blocks.0.mlp.fc2.weight",
             "encoder.blocks.11.mlp.fc2.weight", "encoder.patch_embed.proj.weight", "decoder.conv6.weight", "classifier.weight")
    cos = {n: [] for n in names}
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    for seed in STUDENT_SEEDS:
        args = default_args("VOC12", crop_size=S, teacher_precision="fp16c8", teacher_graph=False, teacher_async=False)
        tr = CoSATrainer(args, dev, seed=seed)
        sd = {k: v.detach().cpu().clone() for k, v in tr.student.state_dict().items()}
        wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=seed + 2)
        n_iter = args.warmup_iters + 1
        loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
        cpu = CpuStep(sd, num_classes=21, aux_layer=-4)
        closs, clogs = cpu.losses(wimg.cpu(), simg.cpu(), lab.cpu(), box.numpy(), n_iter)
        for k in ("cls_loss", "cls_aux_loss", "seg_loss", "cam_loss", "overall_loss"):
            assert float(logs[k]) == pytest.approx(float(clogs[k]), rel=1e-3, abs=2e-5), (seed, k, float(logs[k]), float(clogs[k]))
        tr.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        closs.backward()
        named = dict(tr.student.named_parameters())
        for n in names:
            cos[n].append(_cos(named[n].grad.float().cpu(), cpu.student.p(n).grad))
        del tr, cpu
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r05_student_cosines_over_seeds.txt"), "w") as f:
        f.write(f"# weight-gradient cosines, HIP student (fp32 stream, bf16 MFMA operands) vs oracle/cpu_step.py, ViT-B, S = {S}, b = 2, seeds {STUDENT_SEEDS}\n")
        for n in names:
            f.write(f"{n}: mean {np.mean(cos[n]):.5f} min {np.min(cos[n]):.5f}  " + " ".join(f"{c:.5f}" for c in cos[n]) + "\n")
    for n in names:
        enc = "encoder" in n
        assert np.mean(cos[n]) >= (0.996 if enc else 0.999) and np.min(cos[n]) >= (0.993 if enc else 0.9985), (n, cos[n])          # measured: encoder mean >= 0.9974, min >= 0.9963 (profiles/r05_student_cosines_over_seeds.txt)


def test_ten_step_trajectory_vs_cpu_oracle():
    """Ten optimizer steps of the real trainer (fp32-stream student, parity-grade teacher, fused AdamW + EMA kernel) next to ten steps of
    oracle/cpu_step.py on the same weights and the same batch (S = 224, b = 2, post-warm-up loss weights, the LR schedule from step 0):
    per-step losses within 1e-2 relative, and the accumulated parameter update theta_10 - theta_0 compared by cosine.  AdamW's update is
    m / sqrt(v) -- sign-like in the first steps -- so an element whose gradient is smaller than the rounding noise flips: the update cosine
    is far more sensitive than the gradient cosine (gradient cosine 0.999 = 4.5 % noise -> ~1.4 % flipped signs -> update cosine ~0.97).
    Bars per layer kind below; measured values in profiles/r04_trajectory_vs_oracle.txt."""
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    from oracle.cpu_step import CpuStep
    dev = torch.device("cuda", 0)
    S, b, C, steps = 224, 2, 20, 10
    args = default_args("VOC12", crop_size=S, teacher_precision="fp16c8", teacher_graph=False, teacher_async=False)
    tr = CoSATrainer(args, dev, seed=3)
    sd = {k: v.detach().cpu().clone() for k, v in tr.student.state_dict().items()}
    wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=5)
    cw, cs_, cl, cb = wimg.cpu(), simg.cpu(), lab.cpu(), box.numpy()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    cpu = CpuStep(sd, num_classes=21, aux_layer=-4)
    names = ("encoder.blocks.0.attn.qkv.weight", "encoder.blocks.5.mlp.fc1.weight", "encoder.blocks.11.mlp.fc2.weight",
             "encoder.patch_embed.proj.weight", "decoder.conv6.weight", "classifier.weight", "aux_classifier.weight", "encoder.norm.weight")
    named = dict(tr.student.named_parameters())
    t_named = dict(tr.model_AN.named_parameters())
    th0 = {n: sd[n].double() for n in names}
    gsum_h = {n: torch.zeros_like(sd[n], dtype=torch.float64) for n in names}
    gsum_c = {n: torch.zeros_like(sd[n], dtype=torch.float64) for n in names}
    lines = [f"# ten steps, S = {S}, b = {b}: trainer (HIP, fp32 stream) vs oracle/cpu_step.py"]
    n0 = args.warmup_iters + 1
    for it in range(steps):
        logs = tr.step(wimg, simg, lab, box, n0 + it)
        clogs = cpu.step(cw, cs_, cl, cb, n0 + it)
        for n in names:
            gsum_h[n] += named[n].grad.detach().double().cpu()
            gsum_c[n] += cpu.student.p(n).grad.double()
        row = []
        for k in ("cls_loss", "cls_aux_loss", "seg_loss", "cam_loss", "reg_loss", "overall_loss"):
            a, c = float(logs[k]), float(clogs[k])
            row.append(f"{k} {a:.6f}/{c:.6f}")
            assert a == pytest.approx(c, rel=1e-2, abs=1e-4), (it, k, a, c)
        agree = (logs["mask"].cpu().numpy() == clogs["mask"].numpy()).mean()
        assert agree >= 0.999, (it, agree)
        lines.append(f"step {it}: " + "  ".join(row) + f"  label agreement {agree:.5f}")
    checks = []
    for n in names:
        dh = named[n].detach().double().cpu() - th0[n]
        dc = cpu.student.p(n).detach().double() - th0[n]
        cu, cg = _cos(dh, dc), _cos(gsum_h[n], gsum_c[n])
        ratio = float(dh.norm() / (dc.norm() + 1e-300))
        te = float((t_named[n].detach().double().cpu() - cpu.teacher.p(n).detach().double()).abs().max())
        lines.append(f"{n}: update cosine {cu:.5f} (norm ratio {ratio:.4f}, |update| {float(dc.norm()):.3e}); summed-gradient cosine {cg:.5f}; "
                     f"teacher (EMA) max |diff| {te:.2e}")
        checks.append((n, cu, cg, ratio))
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r04_trajectory_vs_oracle.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    for n, cu, cg, ratio in checks:
        head = n.startswith(("decoder.", "classifier", "aux_classifier"))
        assert cg >= (0.999 if head else 0.995), (n, "summed gradient", cg)
        assert cu >= (0.99 if head else 0.95) and 0.95 <= ratio <= 1.05, (n, "update", cu, ratio)


def test_training_step_loss_and_gradients_are_bit_identical_run_to_run():
    """no float atomics are left on the student's path (round 3: weight-gradient splits meet in a fixed-order reduction, the seg-loss sums
    and gradient cells in 64-bit fixed point): the same weights and the same batch give the same loss and the same gradient bits, every
    parameter, every run -- also a race screen for the three-stage LDS-DMA ring of the weight-gradient kernel"""
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    dev = torch.device("cuda", 0)
    S, b, C = 224, 4, 20
    args = default_args("VOC12", crop_size=S, batch_size=b, teacher_graph=False, teacher_async=False)
    tr = CoSATrainer(args, dev, seed=7)
    wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=9)
    n_iter = args.warmup_iters + 1
    runs = []
    for _ in range(3):
        tr.optimizer.zero_grad(set_to_none=True)
        loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
        loss.backward()
        runs.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in tr.student.named_parameters() if p.grad is not None},
                     {k: logs[k].detach().clone() for k in ("seg_loss", "cam_loss", "reg_loss", "cls_loss")}))
    assert len(runs[0][1]) > 100
    for other in runs[1:]:
        assert torch.equal(other[0], runs[0][0])
        for k, v in runs[0][2].items():
            assert torch.equal(other[2][k], v), k
        for n, g in runs[0][1].items():
            assert torch.equal(other[1][n], g), n


@pytest.mark.gpu
@pytest.mark.parametrize("B,K,H,W", [(2, 21, 64, 96), (3, 81, 32, 32), (1, 5, 448, 448)])
def test_softmax_halfres_matches_torch(B, K, H, W):
    """SoftmaxHalfRes (the drop-in get_energy_loss path, utils/seg_helper.py:199-203, 224) against F.softmax + F.interpolate(0.5, bilinear):
    forward and the gradient w.r.t. the logits."""
    import torch.nn.functional as F
    from cosa_amd.utils import seg_helper
    torch.manual_seed(5)
    x = (torch.randn(B, K, H, W, device="cuda") * 3).requires_grad_(True)
    g = torch.randn(B, K, H // 2, W // 2, device="cuda")
    ref = F.interpolate(F.softmax(x, dim=1), scale_factor=0.5, mode="bilinear", align_corners=False, recompute_scale_factor=True)
    (gref,) = torch.autograd.grad(ref, x, g)
    x2 = x.detach().clone().requires_grad_(True)
    out = seg_helper.SoftmaxHalfRes.apply(x2)
    (gout,) = torch.autograd.grad(out, x2, g)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-8)
    # (dp - sum dp p) cancels for the small entries: absolute floor relative to the largest gradient
    assert torch.allclose(gout, gref, rtol=1e-5, atol=1e-6 * float(gref.abs().max()))


@pytest.mark.gpu
def test_get_energy_loss_fused_softmax_path_equals_layer_path():
    """get_energy_loss with the fused softmax + resize kernel against the same function through torch's softmax and the layer's own
    F.interpolate (scale_factor forced off 0.5 by an equal float that fails the == test is not possible: call the layer directly)."""
    import torch.nn.functional as F
    from cosa_amd.utils import seg_helper
    torch.manual_seed(6)
    b, K, S = 2, 21, 96
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    img = torch.randn(b, 3, S, S, device="cuda")
    box = torch.tensor([[0, S, 0, S], [4, S - 8, 2, S - 2]])
    label = torch.randint(0, K, (b, S, S), device="cuda").float()
    label[0, :10] = 255
    logit = torch.randn(b, K, S, S, device="cuda").requires_grad_(True)
    loss = seg_helper.get_energy_loss(img, logit, label, box, layer)
    (g,) = torch.autograd.grad(loss, logit)
    logit2 = logit.detach().clone().requires_grad_(True)
    mean_t = torch.tensor([123.675, 116.28, 103.53], device="cuda")[None, :, None, None]
    std_t = torch.tensor([58.395, 57.12, 57.375], device="cuda")[None, :, None, None]
    crop = seg_helper._crop_mask_from_boxes(box, b, S, S, logit.device)
    ref = layer(img * std_t + mean_t, F.softmax(logit2, dim=1), crop, label.type(torch.uint8).unsqueeze(1))
    (gref,) = torch.autograd.grad(ref, logit2)
    assert float(loss) == pytest.approx(float(ref), rel=1e-5)
    assert torch.allclose(g, gref, rtol=1e-4, atol=1e-12)


@pytest.mark.gpu
def test_seg_loss_non_finite_input_gives_nan_not_garbage():
    """the 64-bit fixed-point sums of the fused seg loss cannot carry NaN / Inf / out-of-range values: a sticky flag word makes the
    loss and its gradient NaN (what float atomics would have propagated) instead of a finite wrong number"""
    from cosa_amd.utils import seg_helper
    from cosa_amd.train_step import synthetic_batch
    dev = torch.device("cuda", 0)
    B, K, S = 2, 21, 224
    wimg, simg, lab, box = synthetic_batch(B, S, 20, dev, seed=3)
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randint(0, 21, (B, S, S), generator=g).float().to(dev)
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    for poison in (float("nan"), float("inf")):
        seg = torch.randn(B, K, S // 16, S // 16, generator=g).to(dev)
        seg[1, 3, 2, 5] = poison
        seg.requires_grad_(True)
        l_seg, l_reg = seg_helper.fused_seg_and_energy_loss(seg, mk(), mk(), simg, box, layer)
        (l_seg + l_reg).backward()
        assert not math.isfinite(float(l_seg)), poison
        assert torch.isnan(seg.grad).all() or not torch.isfinite(seg.grad).all()
    seg = torch.randn(B, K, S // 16, S // 16, generator=g).to(dev).requires_grad_(True)          # and a clean input right after: flag re-armed
    l_seg, l_reg = seg_helper.fused_seg_and_energy_loss(seg, mk(), mk(), simg, box, layer)
    (l_seg + l_reg).backward()
    assert math.isfinite(float(l_seg)) and torch.isfinite(seg.grad).all()


@pytest.mark.gpu
def test_seg_loss_large_weight_with_few_foreground_pixels_is_not_an_overflow():
    """one foreground pixel per image and a loss weight of 8 give per-add gradient contributions of 0.25 * 8 / 2 = 1.0 -- a
    legitimate value that round 4's |v| < 1 bound of the fixed-point cells turned into NaN.  The gradient must be finite and 8 x the
    weight-1 gradient (the backward is linear in the upstream gradient)."""
    from cosa_amd.utils import seg_helper
    from cosa_amd.train_step import synthetic_batch
    dev = torch.device("cuda", 0)
    B, K, S = 2, 21, 224
    wimg, simg, lab, box = synthetic_batch(B, S, 20, dev, seed=3)
    g = torch.Generator().manual_seed(1)
    few = torch.zeros(B, S, S, device=dev)
    few[:, 100, 120] = 3.0
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    seg0 = torch.randn(B, K, S // 16, S // 16, generator=g).to(dev)
    grads = []
    for wgt in (1.0, 8.0):
        seg = seg0.clone().requires_grad_(True)
        l_seg, _ = seg_helper.fused_seg_and_energy_loss(seg, few, few.clone(), simg, box, layer)
        (l_seg * wgt).backward()
        assert math.isfinite(float(l_seg)) and torch.isfinite(seg.grad).all(), wgt
        grads.append(seg.grad.clone())
    assert torch.allclose(grads[1], 8.0 * grads[0], rtol=1e-5, atol=1e-12)
