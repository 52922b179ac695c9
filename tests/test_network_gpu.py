"""GPU parity of the network path: fused attention kernel, ViT/decoder/CAM heads, and the whole training step."""
import math

import numpy as np
import pytest
import torch

from torch_reference import torch_reference_ops

pytestmark = pytest.mark.gpu


def _ref_attention(qkv, H):
    B, N, _ = qkv.shape
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-1, -2)) * 0.125
    lse = torch.logsumexp(att, -1)
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * 64), lse


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (1, 785, 12), (3, 64, 2), (2, 100, 3), (1, 1765, 2), (1, 1, 1), (1, 129, 1)])
def test_attention_fwd_vs_fp32_reference(B, N, H):
    """tolerance: bf16 inputs/outputs, fp32 softmax -> |err| <= 2e-2 * max|ref| (bf16 has 8 mantissa bits)"""
    from cosa_amd import nn_ops
    torch.manual_seed(N)
    qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 1.5).to(torch.bfloat16)
    out, lse = nn_ops._attn_fwd(qkv, B, N, H)
    ref, lse_ref = _ref_attention(qkv, H)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err
    assert torch.allclose(lse, lse_ref, rtol=1e-4, atol=1e-3)


def test_attention_fwd_sharp_rows():
    """forces the online-softmax rescale: one key dominates late in the sequence"""
    from cosa_amd import nn_ops
    B, N, H = 1, 300, 1
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * 64, device="cuda") * 0.5
    qkv[0, 250, 64:128] = qkv[0, 7, 0:64] * 40          # key 250 aligned with query 7
    qkv = qkv.to(torch.bfloat16)
    out, _ = nn_ops._attn_fwd(qkv, B, N, H)
    ref, _ = _ref_attention(qkv, H)
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3


def test_attention_backward_vs_autograd():
    from cosa_amd import nn_ops
    torch.manual_seed(1)
    B, N, H = 2, 197, 4
    qkv = (torch.randn(B, N, 3 * H * 64, device="cuda")).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, N, H * 64, device="cuda").to(torch.bfloat16)
    nn_ops.attention(qkv, H).backward(go)
    g1 = qkv.grad.float().clone()
    q32 = qkv.detach().float().requires_grad_(True)
    _ref_attention(q32, H)[0].backward(go.float())
    assert (g1 - q32.grad).abs().max().item() <= 3e-2 * q32.grad.abs().max().item()


def _tiny_models(golden, dtype):
    from cosa_amd.models import VITNetwork
    from cosa_amd.models import vit
    from oracle.torch_oracle import load_golden_state
    g = golden("vit_tiny")
    net = VITNetwork.__new__(VITNetwork)
    torch.nn.Module.__init__(net)
    from cosa_amd.models import LargeFOV
    net.num_classes = 7
    net.encoder = vit.VisionTransformer(patch_size=16, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4, qkv_bias=True,
                                        aux_layer=-2, num_classes=10, compute_dtype=dtype)
    net.in_channels = [128] * 4
    net.isgap = False
    net.decoder = LargeFOV(128, 7)
    net.isdecoder_trans = False
    net.classifier = torch.nn.Conv2d(128, 6, 1, bias=False)
    net.aux_classifier = torch.nn.Conv2d(128, 6, 1, bias=False)
    net.compute_dtype = dtype
    net.load_state_dict(load_golden_state(g), strict=True)        # the reference's key names load unchanged
    return net.cuda(), g


def test_network_fp32_mode_vs_reference_golden(golden):
    """HOST-LOGIC check on torch's operators (reference operators of tests/torch_reference.py): the module wiring of VITNetwork -- token
    layout, aux layer, decoder, CAM / classification heads, state-dict names -- reproduces the reference's own forward of a 128-wide toy
    encoder within 1e-3.  Not HIP-vs-oracle evidence: no kernel of this repository covers that shape / fp32 operands (the ViT-B HIP path
    is pinned by test_precision_gpu.py and test_losses_gpu.py)."""
    from cosa_amd import nn_ops, _C
    net, g = _tiny_models(golden, torch.float32)
    with torch.no_grad(), pytest.raises(_C.CosaError):          # without the switch the product refuses to leave the HIP path
        net(torch.from_numpy(g["x"]).cuda())
    with torch.no_grad(), torch_reference_ops():
        out = net(torch.from_numpy(g["x"]).cuda())
    for name, o in zip(["cls", "cls_aux", "x4", "seg", "cam", "cam_aux"], out):
        ref = g[name]
        assert np.abs(o.float().cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), name


def test_network_bf16_mode_vs_reference_golden(golden):
    """HOST-LOGIC check (torch reference operators for the 128-wide toy encoder's projections, HIP attention): bf16 wiring within 3e-2 of
    the output range of the reference's forward"""
    from cosa_amd import nn_ops
    net, g = _tiny_models(golden, torch.bfloat16)
    with torch.no_grad(), torch_reference_ops():
        out = net(torch.from_numpy(g["x"]).cuda())
    for name, o in zip(["cls", "cls_aux", "x4", "seg", "cam", "cam_aux"], out):
        ref = g[name]
        assert np.abs(o.float().cpu().numpy() - ref).max() <= 3e-2 * np.abs(ref).max() + 1e-3, name


def _vit_b_width_model(golden):
    """the product's VITNetwork at ViT-B width, depth 2, with the recipe weights of tests/golden/vit_base_d2.npz"""
    from cosa_amd.models import VITNetwork, LargeFOV
    from cosa_amd.models import vit
    from oracle.gen_golden import recipe_state, VIT_BASE_CFG as cfg
    g = golden("vit_base_d2")
    shapes = {str(k): tuple(int(d) for d in str(s_).split(",")) for k, s_ in zip(g["shape_keys"], g["shape_dims"])}
    sd, digest = recipe_state(shapes)
    assert digest == str(g["weights_sha256"])
    E, C1 = cfg["embed_dim"], cfg["num_classes"]
    net = VITNetwork.__new__(VITNetwork)
    torch.nn.Module.__init__(net)
    net.num_classes = C1
    net.encoder = vit.VisionTransformer(patch_size=16, embed_dim=E, depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=4, qkv_bias=True,
                                        aux_layer=cfg["aux_layer"], num_classes=1000, compute_dtype=torch.bfloat16)
    net.in_channels = [E] * 4
    net.isgap = False
    net.decoder = LargeFOV(E, C1)
    net.isdecoder_trans = False
    net.classifier = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    net.aux_classifier = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    net.compute_dtype = torch.bfloat16
    net.load_state_dict(sd, strict=True)        # the reference's key names load unchanged
    return net.cuda().eval(), g


@pytest.mark.parametrize("mode,tol", [("fp16x3", 1e-4), ("bf16x3", 1e-3), ("fp16c8", 1e-3), ("fp16c8-x1", 1e-3), ("fp16c4", 2e-3), ("bf16", 3e-2)])
def test_hip_network_at_vit_b_width_vs_reference_golden(golden, mode, tol):
    """a REFERENCE-produced vector through the HIP kernels themselves -- the reference's VisionTransformer(embed 768, 12
    heads, depth 2) + LargeFOV + classifiers (models/__init__.py:163-206, models/vit/vit.py:302-321) produced the six outputs of
    tests/golden/vit_base_d2.npz; the product's VITNetwork of that geometry runs its no-grad pass on the persistent MFMA GEMMs, the DMA
    attention, the fused LayerNorms and the exact-fp32 heads (NO torch operator installed: outside `torch_reference_ops` any fallback raises)
    and must reproduce them within `tol` of each output's range (1e-3 for the parity-grade operand modes)."""
    net, g = _vit_b_width_model(golden)
    net.set_nograd_precision(mode)
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        assert net.encoder.use_fused(x), "the fused HIP path must be the one that runs"
        out = net(x)
    for name, o in zip(["cls", "cls_aux", "x4", "seg", "cam", "cam_aux"], out):
        ref = g[name]
        assert tuple(o.shape) == ref.shape, name
        err = np.abs(o.float().cpu().numpy() - ref).max() / np.abs(ref).max()
        # the decoder's convolutions take the final tokens as ONE 16-bit copy in every mode (bf16: 8 significant bits, fp16: 11) -- `seg` is not
        # part of the CAM tolerance; the other five outputs are formed from the fp32 tokens
        bound = tol if name != "seg" else max(tol, 1e-2 if net.compute_dtype == torch.bfloat16 else 2e-3)
        assert err <= bound, (mode, name, err)


def test_state_dict_keys_match_reference(golden):
    from cosa_amd.models import VITNetwork
    g = golden("vit_tiny")
    ref_keys = {k.split("/", 1)[1] for k in g if k.startswith(("sd/", "sdq/"))}
    net = VITNetwork("vit_base_patch16_224", 21, pretrained=False)
    mine = set(net.state_dict().keys())
    strip = lambda ks: {k for k in ks if not k.startswith("encoder.blocks.")} | \
        {k.split(".", 3)[3] for k in ks if k.startswith("encoder.blocks.0.")}
    assert strip(mine) == strip(ref_keys)
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) // 1_000_000 == 92      # voc_log.txt:84


def test_training_step_fp32_vs_cpu_oracle():
    """HOST-LOGIC check: the training step's algebra (teacher multi-scale pass -> label maps -> five losses -> backward) with the NETWORK on
    torch's fp32 operators (test-only mode) and the label / loss kernels on HIP, vs oracle/cpu_step.py on identical weights and inputs: label
    maps agree >= 0.999, losses within 2e-3.  The ViT-B bf16 student on the HIP kernels is compared with the same oracle in
    tests/test_losses_gpu.py."""
    from cosa_amd import nn_ops
    with torch_reference_ops():
        _training_step_fp32_vs_cpu_oracle()


def _training_step_fp32_vs_cpu_oracle():
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    from oracle.cpu_step import CpuStep
    dev = torch.device("cuda", 0)
    S, b, C = 96, 2, 20
    args = default_args("VOC12", crop_size=S, compute_dtype=torch.float32)
    tr = CoSATrainer(args, dev, seed=3)
    sd = {k: v.detach().cpu().clone() for k, v in tr.student.state_dict().items()}
    wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=5)
    n_iter = args.warmup_iters + 1
    loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
    cpu = CpuStep(sd, num_classes=21, aux_layer=-4)
    closs, clogs = cpu.losses(wimg.cpu(), simg.cpu(), lab.cpu(), box.numpy(), n_iter)
    m_gpu, m_cpu = logs["mask"].cpu().numpy(), clogs["mask"].numpy()
    agree = (m_gpu == m_cpu).mean()
    assert agree >= 0.999, f"label agreement {agree}"
    for k in ("cls_loss", "cls_aux_loss", "seg_loss", "cam_loss", "reg_loss", "overall_loss"):
        a, c = float(logs[k]), float(clogs[k])
        assert a == pytest.approx(c, rel=2e-3, abs=1e-7), (k, a, c)
    # gradients flow and the update runs
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    cpu.opt.zero_grad(set_to_none=True)
    closs.backward()
    gq = tr.student.encoder.blocks[0].attn.qkv.weight.grad.cpu()
    cq = cpu.student.p("encoder.blocks.0.attn.qkv.weight").grad
    assert (gq - cq).abs().max().item() <= 2e-2 * cq.abs().max().item() + 1e-9


def test_training_step_bf16_runs_and_learns():
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    dev = torch.device("cuda", 0)
    args = default_args("VOC12", crop_size=128)
    tr = CoSATrainer(args, dev, seed=0)
    wimg, simg, lab, box = synthetic_batch(4, 128, 20, dev, seed=2)
    first = None
    for it in range(6):
        logs = tr.step(wimg, simg, lab, box, n_iter=args.warmup_iters + 1 + it)
        v = float(logs["cls_loss"])
        assert math.isfinite(float(logs["overall_loss"]))
        first = v if first is None else first
    assert v < first                         # same batch six times: the classification loss must go down
    t0 = next(tr.model_AN.parameters()).detach().clone()
    assert set(torch.unique(logs["mask"]).tolist()) <= set(range(21)) | {255}


def test_fused_seg_and_energy_loss_vs_unfused():
    """fused forward/backward kernels vs the op-by-op path (F.interpolate -> seg_loss x2 -> get_energy_loss), fp32:
    losses 1e-4 relative, gradient wrt the low-res logits 2e-3 of its max."""
    import torch.nn.functional as F
    from cosa_amd.utils import seg_helper
    torch.manual_seed(0)
    B, K, hs, S = 3, 21, 12, 192
    seg_lr = (torch.randn(B, K, hs, hs, device="cuda") * 2).requires_grad_(True)
    rng = np.random.default_rng(0)
    mk = lambda: torch.from_numpy(rng.choice([0, 1, 5, 20, 255], size=(B, S, S), p=[0.4, 0.15, 0.15, 0.1, 0.2]).astype(np.float32)).cuda()
    mA, mB = mk(), mk()
    mA[2] = 255                                    # an image with nothing labelled in the main mask
    simg = torch.randn(B, 3, S, S, device="cuda")
    box = torch.tensor([[0, S, 0, S], [10, 150, 20, 190], [0, S, 0, 100]], dtype=torch.int16)
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    w_seg, w_reg = 0.1, 0.05
    # reference path
    up = F.interpolate(seg_lr, size=(S, S), mode="bilinear", align_corners=False)
    l_seg = 0.5 * seg_helper.seg_loss(up, mA) + 0.5 * seg_helper.seg_loss(up, mB)
    l_reg = seg_helper.get_energy_loss(img=simg, logit=up, label=mA, img_box=box, loss_layer=layer)
    (w_seg * l_seg + w_reg * l_reg).sum().backward()
    g_ref = seg_lr.grad.clone()
    seg_lr.grad = None
    f_seg, f_reg = seg_helper.fused_seg_and_energy_loss(seg_lr, mA, mB, simg, box, layer)
    (w_seg * f_seg + w_reg * f_reg).sum().backward()
    assert float(f_seg) == pytest.approx(float(l_seg), rel=1e-4)
    assert float(f_reg) == pytest.approx(float(l_reg), rel=1e-3)
    err = (seg_lr.grad - g_ref).abs().max().item()
    assert err <= 2e-3 * g_ref.abs().max().item(), (err, g_ref.abs().max().item())
    # the same with the lattice of the strong image built ahead on a side stream (PreparedLattice): same lattice, same numbers
    g_fused = seg_lr.grad.clone()
    seg_lr.grad = None
    prep = seg_helper.PreparedLattice(layer.sigma_rgb, layer.sigma_xy * layer.scale_factor)
    prep.start(simg, K)
    p_seg, p_reg = seg_helper.fused_seg_and_energy_loss(seg_lr, mA, mB, simg, box, layer, prepared=prep)
    (w_seg * p_seg + w_reg * p_reg).sum().backward()
    assert prep.key is None                                         # consumed
    assert float(p_seg) == pytest.approx(float(f_seg), rel=1e-6) and float(p_reg) == pytest.approx(float(f_reg), rel=1e-5)
    assert (seg_lr.grad - g_fused).abs().max().item() <= 1e-5 * g_fused.abs().max().item()
    # a lattice prepared for another shape is ignored (falls back to building in place)
    prep.start(simg[:2], K)
    q_seg, q_reg = seg_helper.fused_seg_and_energy_loss(seg_lr, mA, mB, simg, box, layer, prepared=prep)
    assert float(q_reg) == pytest.approx(float(f_reg), rel=1e-6)


def test_cam_loss_targets_vs_unfused():
    """cam_loss targets from per-scale low-res teacher segs == seg_refine_by_label(full-res seg) resized (tolerance: the
    temperature 0.01 amplifies 1e-7 logit noise by 100)."""
    import torch.nn.functional as F
    from cosa_amd.utils import seg_helper
    torch.manual_seed(4)
    B, K, S = 3, 21, 128
    sizes = [8, 4, 12]
    segs = [torch.randn(2 * B, K, n, n, device="cuda") * 0.05 for n in sizes]
    labels = torch.zeros(B, K - 1, device="cuda")
    labels[0, [2, 7]] = 1
    labels[1, [0]] = 1
    full = None
    for t in segs:
        up = F.interpolate(t, size=(S, S), mode="bilinear", align_corners=False)
        v = up[:B] + up[B:].flip(-1)
        full = v if full is None else full + v
    ref = seg_helper.seg_refine_by_label(full, labels, softmaxtemp=0.01)
    ref = F.interpolate(ref[:, 1:], size=(8, 8), mode="bilinear", align_corners=False)
    out = seg_helper.cam_loss_targets(segs, labels, S, (8, 8), 0.01)
    assert torch.allclose(out, ref, atol=2e-4, rtol=1e-3), (out - ref).abs().max().item()
    cam = torch.randn(B, K - 1, 8, 8, device="cuda")
    assert float(seg_helper.cam_loss_from_targets(cam, out)) == pytest.approx(float(seg_helper.cam_loss(cam, seg_helper.seg_refine_by_label(full, labels, 0.01))), rel=1e-4)


def test_fused_adamw_ema_step_vs_torch():
    """one multi-tensor kernel == optimizer.step() + EMA + shadow refresh (fp32 params 1e-6, shadows exact casts)"""
    from cosa_amd import nn_ops
    from cosa_amd.utils import torch_helper
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(33, 70), torch.nn.LayerNorm(70), torch.nn.Linear(70, 5)).cuda()
    sa, ta, sb, tb = mk(), mk(), mk(), mk()
    sb.load_state_dict(sa.state_dict()); tb.load_state_dict(ta.state_dict())
    sa[2].bias.requires_grad = False; sb[2].bias.requires_grad = False            # a frozen tensor: EMA only
    def opt_for(m):
        return torch_helper.PolyWarmupAdamW([{"params": [p for p in m[0].parameters()], "lr": 6e-5, "weight_decay": 1e-2},
                                             {"params": [p for p in list(m[1].parameters()) + [m[2].weight]], "lr": 6e-4, "weight_decay": 1e-2}],
                                            lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=3, max_iter=100, warmup_ratio=1e-6, power=0.9)
    oa, ob = opt_for(sa), opt_for(sb)
    sh_s, sh_t = nn_ops.ShadowSet(sb), nn_ops.ShadowSet(tb)
    fused = torch_helper.FusedAdamWEMAStep(ob, list(sb.parameters()), list(tb.parameters()), 0.9, shadow_of=nn_ops.shadow_of)
    for it in range(5):
        x = torch.randn(16, 33, device="cuda")
        for m, o in ((sa, oa), (sb, ob)):
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
        oa.step()
        torch_helper.ema_update(list(ta.parameters()), list(sa.parameters()), 0.9)
        fused.step()
        assert oa.param_groups[1]["lr"] == pytest.approx(ob.param_groups[1]["lr"], rel=1e-12)
    for a, b in zip(list(sa.parameters()) + list(ta.parameters()), list(sb.parameters()) + list(tb.parameters())):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), (a - b).abs().max().item()
    teacher_ids = {id(p) for p in tb.parameters()}
    for p in list(sb.parameters()) + list(tb.parameters()):
        if p.requires_grad or id(p) in teacher_ids:
            assert torch.equal(nn_ops.shadow_of(p), p.detach().to(torch.bfloat16))


@pytest.mark.parametrize("B,h,w,Cin,Cout", [(2, 28, 28, 768, 512), (3, 9, 14, 128, 128), (1, 5, 5, 64, 256)])
def test_dilated_conv_implicit_gemm_vs_torch(B, h, w, Cin, Cout):
    """LargeFOV conv kernel (NHWC tokens incl. a strided view without the cls row) vs F.conv2d in fp32 on the same bf16 data"""
    import torch.nn.functional as F
    from cosa_amd import nn_ops
    torch.manual_seed(0)
    base = torch.randn(B, h * w + 1, Cin, device="cuda").to(torch.bfloat16)
    tok = base[:, 1:]                                            # strided view, as the encoder hands it over
    wgt = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02).to(torch.bfloat16)
    y = nn_ops.conv3x3_dilated_tokens(tok, wgt, B, h, w, 5, relu=True).view(B, h, w, Cout).permute(0, 3, 1, 2).float()
    x = tok.float().reshape(B, h, w, Cin).permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(x, wgt.float(), padding=5, dilation=5))
    assert (y - ref).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("rows", [(3, 50), (1, 785), (2, 1)])
def test_add_layernorm_fwd_bwd_vs_torch(rows):
    """student blocks' fused residual-add + LayerNorm (cosa_add_layernorm_fwd / cosa_layernorm_bwd) against torch autograd in fp32
    on the same bf16-rounded stream"""
    from cosa_amd import nn_ops
    torch.manual_seed(rows[1])
    B, N = rows
    x = torch.randn(B, N, 768, device="cuda").bfloat16()
    d = (torch.randn(B, N, 768, device="cuda") * 0.5).bfloat16()
    w = torch.nn.Parameter(torch.randn(768, device="cuda") * 0.2 + 1)
    b = torch.nn.Parameter(torch.randn(768, device="cuda") * 0.1)
    gy = torch.randn(B, N, 768, device="cuda").bfloat16()
    gx = torch.randn(B, N, 768, device="cuda").bfloat16()
    for delta in (d, None):
        xs = x.clone().requires_grad_(True)
        ds = delta.clone().requires_grad_(True) if delta is not None else None
        for p in (w, b):
            p.grad = None
        x_new, y = nn_ops.add_layernorm(xs, ds, w, b, 1e-6)
        (y.float() * gy.float()).sum().add((x_new.float() * gx.float()).sum()).backward()
        # reference: fp32 math on the bf16 stream value, bf16 parameters
        xr = x.clone().float().requires_grad_(True)
        dr = delta.clone().float().requires_grad_(True) if delta is not None else None
        wr, br = w.detach().bfloat16().float().requires_grad_(True), b.detach().bfloat16().float().requires_grad_(True)
        sr = (xr + dr).bfloat16().float() if delta is not None else xr
        sr_leaf = sr.detach().requires_grad_(True)
        yr = torch.nn.functional.layer_norm(sr_leaf, (768,), wr, br, 1e-6)
        (yr * gy.float()).sum().add((sr_leaf * gx.float()).sum()).backward()
        assert torch.equal(x_new, sr.bfloat16()) if delta is not None else x_new is xs
        assert (y.float() - yr).abs().max().item() <= 2.0 ** -7 * max(yr.abs().max().item(), 1.0)
        gref = sr_leaf.grad
        tol = 2.0 ** -7 * gref.abs().max().item()
        assert (xs.grad.float() - gref).abs().max().item() <= tol
        if delta is not None:
            assert torch.equal(ds.grad, xs.grad)
        assert (w.grad - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item() + 1e-3
        assert (b.grad - br.grad).abs().max().item() <= 2e-3 * br.grad.abs().max().item() + 1e-3


def test_attention_fwd_dma_ring_is_run_to_run_deterministic():
    """race screen for the LDS-DMA ring of the forward kernel (no atomics: identical bits every run)"""
    from cosa_amd import _C
    torch.manual_seed(3)
    L = _C.lib()
    for (B, N, H) in [(4, 1765, 12), (8, 785, 12), (16, 197, 12), (3, 130, 2)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda").bfloat16()
        ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), qkv.device, "attn")

        def run(flags):
            out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
            lse = torch.empty(B, H, N, device="cuda")
            _C.check(L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, flags, None, _C.ptr(ws), ws.numel(),
                                     _C.stream_ptr()), "cosa_attn_fwd")
            return out, lse
        o0, l0 = run(0)
        for _ in range(20):
            o, l = run(0)
            assert torch.equal(o, o0) and torch.equal(l, l0)


def test_attention_backward_is_run_to_run_deterministic():
    """no atomics in the backward either (dQ and dK/dV are owned by disjoint workgroups): identical bits every run = race screen
    for its LDS-DMA rings and transposing reads"""
    from cosa_amd import nn_ops
    torch.manual_seed(4)
    for (B, N, H) in [(3, 785, 12), (2, 197, 12), (2, 130, 3)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda").bfloat16().requires_grad_(True)
        g = torch.randn(B, N, H * 64, device="cuda").bfloat16()
        first = None
        for _ in range(10):
            qkv.grad = None
            nn_ops.attention(qkv, H).backward(g)
            if first is None:
                first = qkv.grad.clone()
            assert torch.equal(qkv.grad, first)


def test_teacher_pass_flip_equivariance_at_bench_size():
    """size-independent property at BASELINE's configuration (b=16, 448^2, ViT-B bf16: 25 120 token rows through the persistent GEMM,
    the DMA attention kernel, the implicit-GEMM convs and the fused CAM tail): multi_scale_camseg merges the passes over x and
    flip(x) with max / sum, so feeding flip(x) must give exactly the flipped result -- bit for bit at scale 1.0, where no image
    resampling is involved: the same two network evaluations happen with their batch halves (and hence all tile / workgroup
    positions) swapped."""
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args, synthetic_batch
    from cosa_amd.utils import seg_helper
    torch.manual_seed(0)
    args = default_args("VOC12", crop_size=448, batch_size=16)
    model = build_model(args).cuda().eval()
    wimg, _, lab, _ = synthetic_batch(16, 448, 20, torch.device("cuda"), seed=77)
    cam, aux, seg = seg_helper.multi_scale_camseg(model, wimg, [1.0])
    camf, auxf, segf = seg_helper.multi_scale_camseg(model, wimg.flip(-1).contiguous(), [1.0])
    assert torch.isfinite(cam).all() and cam.min() >= 0 and cam.max() <= 1.0
    assert torch.equal(camf, cam.flip(-1)) and torch.equal(auxf, aux.flip(-1)) and torch.equal(segf, seg.flip(-1))
    # with the 0.5x / 1.5x scales the resampled inputs differ in the last bit between x and flip(x): pseudo labels must still agree
    m = seg_helper.cam2mask(wimg, torch.tensor([[0, 448, 0, 448]] * 16), seg_helper.multi_scale_camseg(model, wimg, [1.0, 0.5, 1.5])[0],
                            lab, 0.7, 0.25, _fold_validation=True)
    mf = seg_helper.cam2mask(wimg, torch.tensor([[0, 448, 0, 448]] * 16),
                             seg_helper.multi_scale_camseg(model, wimg.flip(-1).contiguous(), [1.0, 0.5, 1.5])[0], lab, 0.7, 0.25,
                             _fold_validation=True)
    assert (mf == m.flip(-1)).float().mean().item() > 0.99


@pytest.mark.parametrize("flags", [["--steps", "1", "--warmup", "0"], ["--steps", "2", "--warmup", "1", "--usepar", "--usegmm", "--no-secondary"]])
def test_bench_contract_small(flags):
    """bench.py as the driver calls it (own process), tiny configuration, warm-up counts below the graph-capture threshold: prints ONE JSON
    line with the contract's keys (profiling events must not be taken inside the teacher's graph capture)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "2", "--crop", "224", "--no-cpu-baseline"] + flags,
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == int(flags[1]) and d["roofline"]["bound"] in ("mfma", "hbm")
    # the in-run accuracy check (benchmarked teacher mode against the bf16x3 pass on the bench batch) ran and found no gross error
    live = d["accuracy_live"]
    assert "error" not in live and live["label_agreement"] >= 0.999 and live["mask_miou"] >= 0.998 and "gross_error_ok" in live, live
    assert set(d["tolerance_planes"]) == {"planes", "planes_exempt", "planes_failed", "draws"}          # the exemption count is in the line


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_im2col_flip_and_embed_finish_equal_the_torch_expressions(dt):
    """the teacher's token assembly: cosa_im2col_flip = the im2col view of cat(x, x.flip(-1)) in 16 bits (vit.py:254-262 as a GEMM operand,
    seg_helper.py:241-246), cosa_embed_finish = (cat(cls, tok) + pos).float() (vit.py:303-313) -- both bit for bit, ragged sizes included"""
    from cosa_amd import _C
    L = _C.lib()
    code = 1 if dt == torch.bfloat16 else 2
    torch.manual_seed(5)
    for B, H, W in ((3, 64, 96), (2, 224, 224), (1, 16, 16)):
        x = torch.randn(B, 3, H, W, device="cuda")
        p = 16
        h, w = H // p, W // p
        for flips in (1, 2):
            cols = torch.full((flips * B * h * w + 2, 768), 7.0, device="cuda", dtype=dt)        # two canary rows
            _C.check(L.cosa_im2col_flip(_C.ptr(x), _C.ptr(cols), B, 3, H, W, p, flips, code, _C.stream_ptr()), "im2col")
            xx = torch.cat([x, x.flip(-1)], 0) if flips == 2 else x
            ref = xx.to(dt).reshape(flips * B, 3, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(flips * B * h * w, 768)
            assert torch.equal(cols[:-2], ref) and torch.all(cols[-2:] == 7.0)
        n, D = h * w, 768
        tok = torch.randn(B, n, D, device="cuda").to(dt)
        cls = torch.randn(1, 1, D, device="cuda").to(dt)
        pos = torch.randn(1, n + 1, D, device="cuda").to(dt)
        out = torch.full((B * (n + 1) + 1, D), 7.0, device="cuda")
        _C.check(L.cosa_embed_finish(_C.ptr(tok), _C.ptr(cls), _C.ptr(pos), _C.ptr(out), B, n, D, code, _C.stream_ptr()), "embed")
        ref = (torch.cat((cls.expand(B, -1, -1), tok), dim=1) + pos).float().reshape(-1, D)
        assert torch.equal(out[:-1], ref) and torch.all(out[-1] == 7.0)


@pytest.mark.parametrize("shape,relu", [((16, 20), False), ((3, 80), False), ((4, 20, 28, 28), True), ((2, 81, 40, 40), True), ((1, 5, 3, 7), False)])
def test_multilabel_soft_margin_kernel_vs_torch(shape, relu):
    """the fused loss (classification losses main.py:127-128, cam_loss seg_helper.py:593-602) against F.multilabel_soft_margin_loss: value and
    gradient, row-major logits and NCHW maps (every pixel a row), with the ReLU of cam_loss folded in"""
    from cosa_amd.utils import seg_helper
    torch.manual_seed(len(shape) * 7 + shape[1])
    x = (torch.randn(*shape, device="cuda") * 3).requires_grad_(True)
    y = (torch.rand(*shape, device="cuda") < 0.3).float() if len(shape) == 2 else torch.rand(*shape, device="cuda")
    loss = seg_helper.multilabel_soft_margin(x, y, relu=relu)
    (loss * 1.7).backward()
    x2 = x.detach().clone().requires_grad_(True)
    v = torch.relu(x2) if relu else x2
    if len(shape) == 4:
        C = shape[1]
        ref = torch.nn.functional.multilabel_soft_margin_loss(v.permute(0, 2, 3, 1).reshape(-1, C), y.permute(0, 2, 3, 1).reshape(-1, C))
    else:
        ref = torch.nn.functional.multilabel_soft_margin_loss(v, y)
    (ref * 1.7).backward()
    assert loss.item() == pytest.approx(ref.item(), rel=2e-6, abs=1e-7)
    assert (x.grad - x2.grad).abs().max().item() <= 2e-6 * x2.grad.abs().max().item() + 1e-9


@pytest.mark.gpu
def test_attention_fwd_two_and_four_wave_workgroups_are_bit_identical():
    """the forward kernel with 2 and with 4 waves per workgroup (flags bits 9 / 8; the launcher picks by rounds) computes every query row with
    the same instructions on the same K/V tiles: outputs and LSE must be equal bit for bit, also at ragged lengths"""
    from cosa_amd import _C
    torch.manual_seed(8)
    L = _C.lib()
    for (B, N, H) in [(2, 1765, 12), (3, 513, 4), (1, 2049, 2), (5, 1000, 3), (2, 255, 12)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda").bfloat16()
        ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), qkv.device, "attn")
        res = []
        for flags in (0x200, 0x100, 0):
            out = torch.full((B, N, H * 64), float("nan"), device="cuda", dtype=torch.bfloat16)
            lse = torch.full((B, H, N), float("nan"), device="cuda")
            _C.check(L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, flags, None, _C.ptr(ws), ws.numel(), _C.stream_ptr()),
                     "cosa_attn_fwd")
            res.append((out, lse))
        assert not torch.isnan(res[0][0].float()).any() and not torch.isnan(res[0][1]).any()
        for out, lse in res[1:]:
            assert torch.equal(out, res[0][0]) and torch.equal(lse, res[0][1])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (1, 785, 12), (3, 64, 2), (2, 100, 3), (1, 1765, 2), (1, 1, 1), (1, 129, 1), (2, 513, 4)])
def test_attention_fwd_nograd_variant_vs_fp32_reference(dt, B, N, H):
    """flags bit 10 (the teacher's / evaluation's passes): q pre-scaled in the operand type, the running maximum fed through the score MFMAs
    as a 65th contraction index, row sums over the rounded probabilities (attn_kernels.hip: AUGM).  Same bars as the training-side
    kernel for the output (one more rounding of q: 2e-2 bf16 / 3e-3 fp16 of max|ref|); lse within the operand type's resolution of the
    scores; 2 and 4 waves per workgroup bit-identical; identical bits run to run."""
    from cosa_amd import _C
    torch.manual_seed(N + H)
    L = _C.lib()
    f16 = dt == torch.float16
    fwd = L.cosa_attn_fwd_f16 if f16 else L.cosa_attn_fwd
    qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 1.5).to(dt)
    ws = _C.workspace((L.cosa_attn_workspace_bytes_f16 if f16 else L.cosa_attn_workspace_bytes)(B, N, H), qkv.device, "attn")
    res = []
    for flags in (0x400, 0x400 | 0x200, 0x400 | 0x100, 0x400):
        out = torch.full((B, N, H * 64), float("nan"), device="cuda", dtype=dt)
        lse = torch.full((B, H, N), float("nan"), device="cuda")
        _C.check(fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, flags, None, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_fwd")
        res.append((out, lse))
    for out, lse in res[1:]:
        assert torch.equal(out, res[0][0]) and torch.equal(lse, res[0][1])
    out, lse = res[0]
    ref, lse_ref = _ref_attention(qkv.float(), H)
    err = (out.float() - ref).abs().max().item()
    assert err <= (3e-3 if f16 else 2.5e-2) * ref.abs().max().item() + 1e-3, err
    assert (lse - lse_ref).abs().max().item() <= (4e-3 if f16 else 3e-2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_attention_fwd_nograd_variant_moves_its_reference(dt):
    """the reference maximum of a query (kept representable in the operand type) has to follow: a dominant key late in the sequence, scores far
    below zero from the first tile on (m = 0 must not be taken for a maximum), and a row whose maximum keeps growing tile after tile"""
    from cosa_amd import _C
    L = _C.lib()
    f16 = dt == torch.float16
    fwd = L.cosa_attn_fwd_f16 if f16 else L.cosa_attn_fwd
    B, N, H = 1, 300, 1
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * 64, device="cuda") * 0.5
    qkv[0, 250, 64:128] = qkv[0, 7, 0:64] * 40                       # key 250 aligned with query 7
    qkv[0, :, 64:128] += -3.0 * qkv[0, 9, 0:64].sign()                # query 9: every score strongly negative
    for t in range(4):                                                # query 11: its best key gets better in every 64-key tile
        qkv[0, 64 * t + 5, 64:128] += qkv[0, 11, 0:64] * (6.0 * (t + 1))
    qkv = qkv.to(dt)
    ws = _C.workspace((L.cosa_attn_workspace_bytes_f16 if f16 else L.cosa_attn_workspace_bytes)(B, N, H), qkv.device, "attn")
    out = torch.empty(B, N, H * 64, device="cuda", dtype=dt)
    lse = torch.empty(B, H, N, device="cuda")
    _C.check(fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, 0x400, None, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_fwd")
    ref, lse_ref = _ref_attention(qkv.float(), H)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    assert (out.float() - ref).abs().max().item() <= (3e-3 if f16 else 2.5e-2) * ref.abs().max().item() + 1e-3
    assert ((lse - lse_ref).abs() <= (4e-3 if f16 else 3e-2) + 2e-3 * lse_ref.abs()).all()


def test_attention_fwd_nograd_variant_large_logits_slow_drift_bf16():
    """in the bf16 build the reference maximum is an 8-bit value; at |m| >= 2048 (log2 units) its spacing (16) exceeds the deferral
    threshold (2^6), so a row whose maximum drifts up by 6-8 per 64-key tile used to round its new reference back to the old one.  Scores
    of ~ +/-2500 with such a drift: finite output within the bf16 bar of the fp32 reference."""
    from cosa_amd import _C
    L = _C.lib()
    B, N, H = 1, 64 * 9, 1
    torch.manual_seed(5)
    q = torch.full((64,), 14.0)                                       # |q|^2 = 12544; score = q.k / 8: k = c q gives 1568 c, x log2 e = 2262 c
    qkv = torch.zeros(B, N, 3 * 64)
    qkv[0, :, 0:64] = q
    for t in range(9):                                                # per 64-key tile the best score rises by ~7 log2 units (c + 0.0031)
        qkv[0, 64 * t:64 * t + 64, 64:128] = q * (1.10 + 0.0031 * t) + torch.randn(64, 64) * 0.001
    qkv[0, :, 128:192] = torch.randn(N, 64)
    qkv = qkv.bfloat16().cuda()
    ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), qkv.device, "attn")
    out = torch.empty(B, N, H * 64, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device="cuda")
    _C.check(L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, 0x400, None, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_fwd")
    ref, lse_ref = _ref_attention(qkv.float(), H)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    assert (out.float() - ref).abs().max().item() <= 2.5e-2 * ref.abs().max().item() + 1e-3
    # (the scores themselves carry the 8-bit rounding of the pre-scaled q at magnitude ~2500: lse within that resolution)
    assert ((lse - lse_ref).abs() <= 2.0e-2 * lse_ref.abs()).all()


# Rounds 3-4 pinned the training-side attention kernels by md5 so that ONE draw of the student's bf16 rounding noise stayed comparable from
# round to round; that froze the kernels.  What the pins really screened for -- races and uninitialised reads -- is a
# property of one build: the same input twice must give the same bits (no atomics anywhere in these kernels).  Accuracy is asserted against
# fp32 above (test_attention_fwd_vs_fp32_reference / test_attention_backward_vs_autograd) and, for the whole student, as a multi-seed statistic against the
# CPU oracle (tests/test_losses_gpu.py).
@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 513, 4), (2, 1765, 3)])
def test_attention_fwd_training_kernel_is_deterministic(B, N, H):
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(1234 + N)
    qkv = (torch.randn(B, N, 3 * H * 64, generator=g) * 0.8).bfloat16().cuda()
    out, lse = nn_ops._attn_fwd(qkv, B, N, H)
    for _ in range(3):
        o2, l2 = nn_ops._attn_fwd(qkv, B, N, H)
        assert torch.equal(o2, out) and torch.equal(l2, lse)


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (3, 785, 4), (2, 130, 3)])
def test_attention_backward_is_deterministic(B, N, H):
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(99 + N)
    q0 = (torch.randn(B, N, 3 * H * 64, generator=g) * 0.8).bfloat16().cuda()
    go = (torch.randn(B, N, H * 64, generator=g) * 0.01).bfloat16().cuda()
    grads = []
    for _ in range(3):
        qkv = q0.clone().requires_grad_(True)
        nn_ops.attention(qkv, H).backward(go)
        grads.append(qkv.grad)
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2]) and torch.isfinite(grads[0].float()).all()


@pytest.mark.parametrize("B,N,H", [(8, 1765, 12), (16, 785, 12)])
def test_attention_fwd_nograd_c8_variant_is_deterministic_on_a_full_chip(B, N, H):
    """the teacher's attention launch (fp16 q / k / v, c8 rows out) with every CU busy, twice: the same bytes.  Round 5 found the softmax's
    `v_max3_f32` as inline asm on MFMA accumulators -- outside the compiler's MFMA -> VALU hazard handling: scheduling variants of the kernel
    were run-to-run NON-deterministic at exactly these launch sizes while small launches stayed clean (profiles/r05_attn_variants.txt)."""
    import ctypes
    from cosa_amd import _C
    L = _C.lib()
    g = torch.Generator().manual_seed(7 + N)
    qkv = (torch.randn(B, N, 3 * H * 64, generator=g) * 1.5).half().cuda()
    outs = [torch.zeros(B * N, 4 * H * 64 + 128, device="cuda", dtype=torch.uint8) for _ in range(3)]
    for o in outs:
        _C.check(L.cosa_attn_fwd_f16c8(_C.ptr(qkv), _C.ptr(o), None, B, N, H, 64, ctypes.c_float(0.125), None, _C.stream_ptr()), "cosa_attn_fwd_f16c8")
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
