"""GPU parity of the operand-precision modes of the no-grad (teacher / evaluation) path.

  bf16    8 significant bits per MFMA operand  (the training path's type)
  fp16    11 significant bits, same kernels built with fp16 operands (csrc/op16.hpp)
  bf16x3  16 significant bits: every operand as hi + lo bf16 halves, three MFMA terms (hi*hi + lo*hi + hi*lo), fp32 accumulation
  fp16x3  the same three terms with fp16 halves: 22 significant bits (round 6)
  fp16c8  fp16 hi (11 bits) + two e5m2 correction terms on the block-scaled 8-bit MFMA (~14 bits at 2x the 16-bit work; attention
          operands plain fp16, attention output as c8 rows): the benchmarked parity-grade mode since round 3

Kernel-level checks against fp32 torch with the tolerance of each mode written in the test, then the whole multi-scale teacher pass
(fused ViT-B, embed 768: persistent GEMM + DMA attention + fp32 residual) against oracle/torch_oracle.py on identical weights and
inputs: normalised-CAM relative error, label agreement and mask IoU (BASELINE.json north_star: 1e-3 / bit-exact labels / IoU >= 0.999)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref_gemm(x, w, b, epi, r):
    y = x.float() @ w.float().t() + b.float()
    if epi == 1:
        y = torch.nn.functional.gelu(y)
    if epi == 2:
        y = y + r
    return y


@pytest.mark.parametrize("variant", [0, 1, 6])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_f16_vs_fp32(variant, epi):
    """fp16 operands, fp32 accumulation: output rounding 2^-12 relative -> tolerance 2^-10 of the output scale (fp32 out: 1e-5)"""
    from cosa_amd import nn_ops, _C
    torch.manual_seed(variant * 10 + epi)
    try:
        for (M, N, K) in [(4099, 768, 768), (4608, 2304, 768), (5000, 768, 3072), (300, 768, 768), (1, 128, 64), (66000, 256, 192)]:
            x = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
            b = torch.randn(N, device="cuda").half()
            r = torch.randn(M, N, device="cuda") if epi == 2 else None
            _C.lib().cosa_gemm_set_variant_f16(variant)
            y = nn_ops.gemm_bf16(x, w, b, epi, residual=r)
            ref = _ref_gemm(x, w, b, epi, r)
            assert y.dtype == (torch.float32 if epi == 2 else torch.float16)
            tol = (1e-5 if epi == 2 else 2.0 ** -10) * max(ref.abs().max().item(), 1.0)
            err = (y.float() - ref).abs().max().item()
            assert err <= tol, (variant, epi, M, N, K, err, tol)
    finally:
        _C.lib().cosa_gemm_set_variant_f16(0)


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (1, 785, 12), (2, 100, 3), (1, 1765, 2), (1, 1, 1), (1, 129, 1)])
def test_attention_fwd_f16_vs_fp32(B, N, H):
    """fp16 q/k/v/P/out, fp32 softmax statistics: |err| <= 3e-3 * max|ref| (bf16 build: 2e-2)"""
    from cosa_amd import nn_ops
    torch.manual_seed(N)
    qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 1.5).half()
    out, lse = nn_ops._attn_fwd(qkv, B, N, H)
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-1, -2)) * 0.125
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * 64)
    assert out.dtype == torch.float16
    err = (out.float() - ref).abs().max().item()
    assert err <= 3e-3 * ref.abs().max().item() + 2e-4, err
    assert torch.allclose(lse, torch.logsumexp(att, -1), rtol=1e-4, atol=1e-3)


def test_layernorm_conv_head_f16_vs_torch():
    from cosa_amd import nn_ops
    torch.manual_seed(0)
    x = torch.randn(1000, 768, device="cuda") * 3 + 1
    g, b = (torch.rand(768, device="cuda") + 0.5).half(), torch.randn(768, device="cuda").half()
    y16, y32 = nn_ops.layernorm_f32(x, g, b, 1e-6, True, True)
    ref = torch.nn.functional.layer_norm(x, (768,), g.float(), b.float(), 1e-6)
    assert y16.dtype == torch.float16 and (y32 - ref).abs().max().item() < 1e-4
    assert (y16.float() - ref).abs().max().item() <= 2.0 ** -10 * ref.abs().max().item()
    B, h, w, Cin, Cout = 2, 14, 14, 768, 512
    tok = torch.randn(B, h * w, Cin, device="cuda").half()
    wt = (torch.randn(Cout, Cin, 3, 3, device="cuda") * (9 * Cin) ** -0.5).half()
    y = nn_ops.conv3x3_dilated_tokens(tok, wt, B, h, w, 5, relu=True)
    refc = torch.relu(torch.nn.functional.conv2d(tok.float().view(B, h, w, Cin).permute(0, 3, 1, 2), wt.float(), padding=5, dilation=5))
    refc = refc.permute(0, 2, 3, 1).reshape(B * h * w, Cout)
    assert (y.float() - refc).abs().max().item() <= 2.0 ** -9 * refc.abs().max().item()
    wh = (torch.randn(21, 512, device="cuda") * 0.05).half()
    yh = nn_ops.head_linear(y.view(B, h * w, Cout), wh, round_bf16=True)
    refh = y.float() @ wh.float().t()
    assert (yh - refh).abs().max().item() <= 2.0 ** -10 * refh.abs().max().item() + 1e-6


def _split_ref(v):
    hi = v.bfloat16()
    lo = (v - hi.float()).bfloat16()
    return hi.float() + lo.float()


# halves' dtype -> (GEMM tolerance, attention tolerance, LayerNorm re-composition tolerance) relative to the output scale
_X3 = {torch.bfloat16: (2.0 ** -14, 2.0 ** -13, 2.0 ** -16), torch.float16: (2.0 ** -18, 2.0 ** -17, 2.0 ** -21)}


@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_bf16x3_vs_fp64(epi, hdt):
    """hi + lo bf16 operands, three MFMA terms: the operands carry 16 significant bits (2^-17 relative rounding), the dropped lo*lo
    term is 2^-18; against an fp64 product of the fp32 inputs the error stays below 2^-14 of the output scale, ~100x below bf16.
    Covers the persistent 256x256 kernel (M >= 4096, several jobs per workgroup, M tails) and the 128x128 kernel.
    fp16 halves (round 6, "fp16x3"): 11 + 11 significant bits, lo halves below 2^-14 as fp16 subnormals (the MFMA must NOT flush them:
    this tolerance -- 2^-19, 32x tighter -- is what shows it), the dropped lo*lo term 2^-24."""
    from cosa_amd import nn_ops
    torch.manual_seed(epi)
    for (M, N, K) in [(4099, 768, 768), (4608, 2304, 768), (5000, 768, 3072), (300, 768, 768), (70000, 256, 128), (131, 128, 64)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if epi == 2 else None
        xs, ws = nn_ops.split_rows(x, ones=True, dtype=hdt), nn_ops.split_rows(w, bias=b, dtype=hdt)
        assert xs.dtype == hdt and xs.shape == (M, 2 * K + 64) and torch.equal(xs[:, 2 * K:2 * K + 3].float().cpu(), torch.tensor([1., 1., 0.]).expand(M, 3))
        y = nn_ops.gemm_x3(xs, ws, M, N, K, epi, residual=r.clone() if r is not None else None, ldy=2 * N + 64 if epi != 2 else None)
        ref = x.double() @ w.double().t() + b.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 2:
            ref = ref + r.double()
            got = y.double()
        else:
            assert y.shape == (M, 2 * N + 64) and y.dtype == hdt
            got = y[:, :N].double() + y[:, N:2 * N].double()
        scale = max(ref.abs().max().item(), 1.0)
        err = (got - ref).abs().max().item()
        assert err <= _X3[hdt][0] * scale, (epi, M, N, K, err / scale)


@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_x3_tail_launch_is_interchangeable_with_persistent_jobs(epi, hdt):
    """round 6: the three-term GEMM's leftover jobs (N = 768 at the teacher's M: 1032 = 4 x 256 + 8) run as 128 x 128 quarters on the two-stage
    kernel instead of as a fifth round of the persistent one (launch_v6).  M = 66 000, N = 768: 774 jobs = 3 rounds + 6.  A token's result must
    not depend on which kernel computed it: rolling the rows of X by half the matrix (every row changes tile, most change kernel) rolls the
    output and nothing else, bit for bit -- split 16-bit outputs (two epilogue tiles in LDS) and the fp32 residual form."""
    from cosa_amd import nn_ops
    torch.manual_seed(epi)
    M, N, K = 66000, 768, 768
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    sh = 33000 + 128
    ws = nn_ops.split_rows(w, bias=b, dtype=hdt)
    run = lambda xx, rr: nn_ops.gemm_x3(nn_ops.split_rows(xx, ones=True, dtype=hdt), ws, M, N, K, epi, residual=rr.clone() if rr is not None else None,
                                        ldy=2 * N + 64 if epi != 2 else None)
    y = run(x, r)
    y2 = run(x.roll(sh, 0).contiguous(), r.roll(sh, 0).contiguous() if r is not None else None)
    assert torch.equal(y2[:, :2 * N] if epi != 2 else y2, (y[:, :2 * N] if epi != 2 else y).roll(sh, 0))
    ref = x.double() @ w.double().t() + b.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    got = y.double() if epi == 2 else y[:, :N].double() + y[:, N:2 * N].double()
    if epi == 2:
        ref = ref + r.double()
    assert (got - ref).abs().max().item() <= _X3[hdt][0] * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (1, 785, 3), (2, 100, 3), (1, 1765, 2), (1, 1, 1), (1, 129, 1)])
def test_attention_bf16x3_vs_fp64(B, N, H, hdt):
    from cosa_amd import nn_ops
    torch.manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda") * 1.5
    qs = nn_ops.split_rows(qkv, dtype=hdt)[:, :6 * H * 64].contiguous()      # [hi | lo] rows, as the split qkv projection writes them
    out = torch.zeros(B * N, 2 * H * 64 + 64, device="cuda", dtype=hdt)
    lse = torch.empty(B, H, N, device="cuda")
    nn_ops.attn_fwd_x3(qs, B, N, H, out, lse)
    q, k, v = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-1, -2)) * 0.125
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B * N, H * 64)
    got = out[:, :H * 64].double() + out[:, H * 64:2 * H * 64].double()
    assert (got - ref).abs().max().item() <= _X3[hdt][1] * ref.abs().max().item() + 1e-6
    assert torch.allclose(lse.double(), torch.logsumexp(att, -1), rtol=1e-5, atol=1e-4)          # (fp16 halves: the 2^10 carried by the probabilities is taken out again)
    aug = out[:, 2 * H * 64:].float().cpu()
    assert torch.equal(aug[:, :2], torch.ones(B * N, 2)) and aug[:, 2:].abs().max().item() == 0


@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16])
def test_layernorm_split_vs_torch(hdt):
    from cosa_amd import nn_ops
    torch.manual_seed(0)
    x = torch.randn(1000, 768, device="cuda") * 3 + 1
    g, b = torch.rand(768, device="cuda") + 0.5, torch.randn(768, device="cuda")
    out = torch.empty(1000, 1600, device="cuda", dtype=hdt)
    _, y32 = nn_ops.layernorm_split(x, g, b, 1e-6, out=out, want_f32=True)
    ref = torch.nn.functional.layer_norm(x, (768,), g, b, 1e-6)
    assert (y32 - ref).abs().max().item() < 1e-4
    got = out[:, :768].float() + out[:, 768:1536].float()
    assert (got - y32).abs().max().item() <= _X3[hdt][2] * y32.abs().max().item()
    assert torch.equal(out[:, 1536:1538].float().cpu(), torch.ones(1000, 2)) and out[:, 1538:].float().abs().max().item() == 0


# ---- fp16c8: fp16 hi + e5m2 correction bytes (csrc/c8.hpp) ----------------------------------------------------------------------
def _c8_fields(rows, K):
    """c8 rows [R, 2K + 64] fp16 -> (hi fp32 [R,K], lo8 decoded and unscaled [R,K], hi8 decoded [R,K], aug fp32 [R,64])"""
    R = rows.shape[0]
    raw = rows.contiguous().view(torch.uint8).view(R, 4 * K + 128)
    hi = raw[:, :2 * K].contiguous().view(torch.float16).float()
    lo8 = raw[:, 2 * K:3 * K].contiguous().view(torch.float8_e5m2).float() / 2048.0
    hi8 = raw[:, 3 * K:4 * K].contiguous().view(torch.float8_e5m2).float()
    aug = raw[:, 4 * K:].contiguous().view(torch.float16).float()
    return hi, lo8, hi8, aug


def _c8_ref_fields(v):
    """the same three fields from fp32 values by torch's own conversions (round to nearest even, saturating at the largest finite e5m2)"""
    hi = v.half().float()
    q = lambda t: t.clamp(-57344, 57344).to(torch.float8_e5m2).float()
    return hi, q((v - hi) * 2048.0) / 2048.0, q(hi)


def _c8_emulated_product(x, w, b):
    """fp64 value of what the c8 GEMM computes from fp32 inputs: x_hi w_hi + x_lo8 w_hi8 + x_hi8 w_lo8 + (b_hi + b_lo)"""
    xh, xl, xh8 = (t.double() for t in _c8_ref_fields(x))
    wh, wl, wh8 = (t.double() for t in _c8_ref_fields(w))
    bh = b.half().float()
    bl = (b - bh).half().float()
    return xh @ wh.t() + xl @ wh8.t() + xh8 @ wl.t() + (bh.double() + bl.double())


def test_c8_rows_and_layernorm_c8_fields():
    """producers of c8 rows: every field bit-identical to torch's conversions of the same fp32 values, the value hi + lo8 within
    2^-14 of the input (11 + 3 bits), the augmentation block as specified"""
    from cosa_amd import nn_ops
    torch.manual_seed(0)
    R, K = 777, 768
    v = torch.randn(R, K, device="cuda") * torch.logspace(-3, 3, R, device="cuda")[:, None]
    bias = torch.randn(R, device="cuda")
    rows = nn_ops.c8_rows(v, bias=bias)
    assert rows.shape == (R, 2 * K + 64) and rows.dtype == torch.float16
    hi, lo8, hi8, aug = _c8_fields(rows, K)
    rh, rl, rh8 = _c8_ref_fields(v)
    assert torch.equal(hi, rh) and torch.equal(lo8, rl) and torch.equal(hi8, rh8)
    assert ((hi + lo8 - v).abs() <= 2.0 ** -14 * v.abs() + 2.0 ** -27).all()          # (absolute floor: fp16 subnormals)
    bh = bias.half().float()
    assert torch.equal(aug[:, 0], bh) and torch.equal(aug[:, 1], (bias - bh).half().float()) and aug[:, 2:].abs().max().item() == 0
    ones = _c8_fields(nn_ops.c8_rows(v, ones=True), K)[3]
    assert torch.equal(ones[:, :2].cpu(), torch.ones(R, 2)) and ones[:, 2:].abs().max().item() == 0
    # saturation: values beyond the largest finite e5m2 (57344) keep a finite hi8
    big = torch.full((4, 128), 65000.0, device="cuda")
    assert torch.equal(_c8_fields(nn_ops.c8_rows(big), 128)[2], torch.full((4, 128), 57344.0, device="cuda"))
    # round 6: the VALUE saturates once (c8.hpp:c8_sat) -- beyond +-57344, fp16's own range included, every field is the one of +-57344 (hi finite,
    # lo8 = 0); up to it the fields are torch's conversions bit for bit (first part of this test: magnitudes up to ~4e3; here the top of the range)
    huge = torch.tensor([1e6, -1e6, 65504.0, -70000.0, float("inf"), 57344.0, -57344.0, 3e38], device="cuda").repeat(16)[None].repeat(4, 1)
    hh, hl, hh8, _ = _c8_fields(nn_ops.c8_rows(huge), 128)
    assert torch.equal(hh, huge.clamp(-57344, 57344)) and hl.abs().max().item() == 0 and torch.equal(hh8, huge.clamp(-57344, 57344))
    top = (torch.rand(4, 128, device="cuda") * 2 - 1) * 57344.0
    th, tl, th8, _ = _c8_fields(nn_ops.c8_rows(top), 128)
    rh, rl, rh8 = _c8_ref_fields(top)
    assert torch.equal(th, rh) and torch.equal(tl, rl) and torch.equal(th8, rh8)
    x = torch.randn(1000, 768, device="cuda") * 3 + 1
    g, b = torch.rand(768, device="cuda") + 0.5, torch.randn(768, device="cuda")
    out = torch.empty(1000, 1600, device="cuda", dtype=torch.float16)
    _, y32 = nn_ops.layernorm_c8(x, g, b, 1e-6, out=out, want_f32=True)
    ref = torch.nn.functional.layer_norm(x, (768,), g, b, 1e-6)
    assert (y32 - ref).abs().max().item() < 1e-4
    hi, lo8, hi8, aug = _c8_fields(out, 768)
    rh, rl, rh8 = _c8_ref_fields(y32)
    assert torch.equal(hi, rh) and torch.equal(lo8, rl) and torch.equal(hi8, rh8)
    assert torch.equal(aug[:, :2].cpu(), torch.ones(1000, 2)) and aug[:, 2:].abs().max().item() == 0


@pytest.mark.parametrize("hdt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_im2col_split_tokens_equal_split_rows_of_the_im2col_view(hdt):
    """cosa_im2col_flip_split_tokens[_f16] (round 6: the x3 teacher's patch rows written into a token-shaped operand) = split_rows(ones) of the
    im2col view of cat(x, x.flip(-1)), bit for bit, at the rows behind every image's class-token slot; the slots and the canary row stay untouched"""
    from cosa_amd import nn_ops, _C
    torch.manual_seed(6)
    p = 16
    for B, H, W in ((3, 64, 96), (2, 224, 224), (1, 16, 16)):
        x = torch.randn(B, 3, H, W, device="cuda") * 2
        h, w = H // p, W // p
        n = h * w
        for flips in (1, 2):
            for cls_rows in (1, 0):
                rows = torch.full((flips * B * (n + cls_rows) + 1, nn_ops.split_ld(768)), 7.0, device="cuda", dtype=hdt)      # one canary row
                _C.check(nn_ops._x3_fn("cosa_im2col_flip_split_tokens", hdt)(_C.ptr(x), _C.ptr(rows), B, 3, H, W, p, flips, cls_rows, _C.stream_ptr()), "im2col")
                xx = torch.cat([x, x.flip(-1)], 0) if flips == 2 else x
                cols = xx.reshape(flips * B, 3, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(flips * B * n, 768).contiguous()
                want = nn_ops.split_rows(cols, ones=True, dtype=hdt).view(flips * B, n, -1)
                got = rows[:-1].view(flips * B, n + cls_rows, -1)
                assert torch.equal(got[:, cls_rows:].contiguous().view(torch.int16), want.contiguous().view(torch.int16))
                assert torch.all(got[:, :cls_rows] == 7.0) and torch.all(rows[-1] == 7.0)


def test_im2col_c8_and_batched_weight_rows_equal_the_row_producer():
    """cosa_im2col_flip(dtype 3) = c8_rows of the im2col view of cat(x, x.flip(-1)); cosa_c8_rows_batched (all weight matrices of a network in
    one launch) = c8_rows per matrix -- both bit for bit"""
    from cosa_amd import nn_ops, _C
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    L = _C.lib()
    torch.manual_seed(5)
    for B, H, W in ((3, 64, 96), (2, 224, 224), (1, 16, 16)):
        x = torch.randn(B, 3, H, W, device="cuda") * 2
        p = 16
        h, w = H // p, W // p
        for flips in (1, 2):
            rows = torch.full((flips * B * h * w + 1, nn_ops.split_ld(768)), 7.0, device="cuda", dtype=torch.float16)      # one canary row
            _C.check(L.cosa_im2col_flip(_C.ptr(x), _C.ptr(rows), B, 3, H, W, p, flips, 3, _C.stream_ptr()), "im2col")
            xx = torch.cat([x, x.flip(-1)], 0) if flips == 2 else x
            cols = xx.reshape(flips * B, 3, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(flips * B * h * w, 768).contiguous()
            assert torch.equal(rows[:-1].view(torch.int16), nn_ops.c8_rows(cols, ones=True).view(torch.int16)) and torch.all(rows[-1] == 7.0)
    net = build_model(default_args("VOC12", crop_size=64)).cuda().eval()
    net.set_nograd_precision("fp16c8-9")
    with torch.no_grad():
        for q in net.parameters():
            q.add_(torch.randn_like(q) * 0.01)                                        # (biases are zero-initialised)
    ws = net.encoder._c8_weights()
    assert len(ws) == 1 + 4 * 9 and "9.qkv" not in ws
    blk = net.encoder.blocks[3]
    for name, wgt, b in (("patch", net.encoder.patch_embed.proj.weight.reshape(768, -1), net.encoder.patch_embed.proj.bias),
                         ("3.qkv", blk.attn.qkv.weight, blk.attn.qkv.bias), ("3.fc2", blk.mlp.fc2.weight, blk.mlp.fc2.bias)):
        wgt, b = wgt.detach().clone().contiguous(), b.detach().clone()
        if name.endswith(".qkv"):          # the attention scale 64^-0.5 log2(e) is folded into the q rows (the attention kernel is called with ln 2)
            c = torch.tensor(0.125, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
            wgt[:768] *= c
            b[:768] *= c
        assert torch.equal(ws[name].view(torch.int16), nn_ops.c8_rows(wgt, bias=b).view(torch.int16)), name


@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_f16c8_vs_fp64(epi):
    """fp16 x fp16 on the 16-bit MFMA + two e5m2 correction terms on the block-scaled MFMA (K = 128 per instruction, E8M0 scale 2^-11).
    (a) against the fp64 value of exactly those four terms: 5e-6 of the output scale (fp32 accumulation only) -- this is the check of
    the tile / lane / scale plumbing; (b) against the fp64 product of the fp32 inputs: 2^-13 (fp16 alone: 2^-11).  Shapes cover one and
    several jobs per persistent workgroup, M tails, M < 256 and both K of the network."""
    from cosa_amd import nn_ops
    torch.manual_seed(epi)
    for (M, N, K) in [(300, 768, 768), (4608, 2304, 768), (5000, 768, 3072), (100, 3072, 768), (66000, 256, 256), (1, 256, 128)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if epi == 2 else None
        xs, ws = nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b)
        y = nn_ops.gemm_c8(xs, ws, M, N, K, epi, residual=r.clone() if r is not None else None)
        emu = _c8_emulated_product(x, w, b)
        ref = x.double() @ w.double().t() + b.double()
        if epi == 1:
            emu, ref = torch.nn.functional.gelu(emu), torch.nn.functional.gelu(ref)
        if epi == 2:
            emu, ref = emu + r.double(), ref + r.double()
        scale = max(ref.abs().max().item(), 1.0)
        if epi == 2:
            assert y.dtype == torch.float32 and y.shape == (M, N)
            got = y.double()
        elif epi == 0:
            assert y.dtype == torch.float16 and y.shape == (M, N)
            got = y.double()
        else:
            assert y.dtype == torch.float16 and y.shape == (M, 2 * N + 64)
            hi, lo8, hi8, _ = _c8_fields(y, N)
            assert torch.equal(hi8, hi.clamp(-57344, 57344).to(torch.float8_e5m2).float()), "hi8 must be the e5m2 rounding of the stored hi"
            got = hi.double() + lo8.double()
        tol_emu = (2.0 ** -11 if epi == 0 else (2.0 ** -13 if epi == 1 else 5e-6)) * scale       # epi 0 / 1 add the output's own rounding
        assert (got - emu).abs().max().item() <= tol_emu, (epi, M, N, K, (got - emu).abs().max().item() / scale)
        tol = (2.0 ** -11 if epi == 0 else 2.0 ** -13) * scale
        assert (got - ref).abs().max().item() <= tol, (epi, M, N, K, (got - ref).abs().max().item() / scale)


def test_gemm_f16c8_residual_beats_fp16_by_its_correction_terms():
    """the correction terms are doing their job: on the same inputs the c8 residual GEMM is >= 6x closer to fp64 than the plain fp16 GEMM"""
    from cosa_amd import nn_ops
    torch.manual_seed(11)
    M, N, K = 4608, 768, 3072
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    r = torch.zeros(M, N, device="cuda")
    ref = x.double() @ w.double().t() + b.double()
    y8 = nn_ops.gemm_c8(nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b), M, N, K, 2, residual=r.clone())
    y16 = nn_ops.gemm_bf16(x.half(), w.half(), b.half(), 2, residual=r.clone())
    e8, e16 = (y8.double() - ref).abs().mean().item(), (y16.double() - ref).abs().mean().item()
    assert e8 * 6 <= e16, (e8, e16)


@pytest.mark.parametrize("B,N,H", [(2, 197, 12), (1, 785, 3), (2, 100, 3), (1, 1765, 2), (1, 1, 1), (1, 129, 1)])
def test_attention_c8_output_rows(B, N, H):
    """fp16 attention whose output leaves as c8 rows: hi is the plain fp16 kernel's output (up to double rounding), hi + lo8 is the fp32 result
    before that rounding (so it is closer to the fp32 reference of the same fp16 inputs), hi8 / aug as specified"""
    from cosa_amd import nn_ops
    torch.manual_seed(N)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D, device="cuda") * 1.5).half()
    plain, lse_p = nn_ops._attn_fwd(qkv, B, N, H, nograd=True)          # (the same no-grad variant of the kernel as the c8-output one)
    out = torch.zeros(B * N, 2 * D + 64, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, N, device="cuda")
    nn_ops.attn_fwd_c8(qkv, B, N, H, out, lse)
    hi, lo8, hi8, aug = _c8_fields(out, D)
    # (the plain kernel's o * (1 / l) is rounded to fp16 ONCE by a mixed-precision fma, here the fp32 product is kept for lo8 and rounded
    # again: the two agree except for rare double-rounding cases, which are one fp16 ulp apart)
    p = plain.view(B * N, D).float()
    assert ((hi - p).abs() <= 2.0 ** -10 * p.abs() + 1e-7).all() and (hi != p).float().mean().item() < 1e-3 and torch.equal(lse, lse_p)
    assert torch.equal(hi8, hi.clamp(-57344, 57344).to(torch.float8_e5m2).float())
    assert torch.equal(aug[:, :2].cpu(), torch.ones(B * N, 2)) and aug[:, 2:].abs().max().item() == 0
    q, k, v = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, D)
    e_c8, e_hi = (hi.double() + lo8.double() - ref).abs().max().item(), (hi.double() - ref).abs().max().item()
    assert e_c8 <= 3e-3 * ref.abs().max().item() + 2e-4 and e_c8 <= e_hi * 1.05 + 1e-7


def test_no_grad_forward_sees_weights_written_through_data_and_raw_pointers():
    """the reference loop's EMA is `param.data.mul_(m).add_(...)`, which does not bump `_version`; the 16-bit weights a
    no-grad forward reads must follow it all the same (nothing is cached by version)."""
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    torch.manual_seed(0)
    net = build_model(default_args("VOC12", crop_size=64)).cuda().eval()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    with torch.no_grad():
        a = net(x)[4].clone()
        for p in net.parameters():
            p.data.mul_(0.5).add_(0.01)
        b = net(x)[4].clone()
        c = net(x)[4].clone()
    assert not torch.allclose(a, b), "forward after an in-place .data update still used the old 16-bit weights"
    assert torch.equal(b, c)


# ---- the whole teacher pass vs the fp32 CPU oracle ----------------------------------------------------------------------------
def _miou(a, b, n=21):
    ious = []
    for c in list(range(n)) + [255]:
        A, B = a == c, b == c
        u = (A | B).sum()
        if u:
            ious.append((A & B).sum() / u)
    return float(np.mean(ious))


_ORACLE = {}


def teacher_csrc_hash():
    """sha256 over the kernel sources the teacher's no-grad pass and cam2mask run through: the accuracy record carries it, and bench.py
    refuses to read `tolerance_met` from a record taken with other kernels"""
    import hashlib
    h = hashlib.sha256()
    for f in TEACHER_CSRC:
        h.update(open(os.path.join(ROOT, "cosa_amd", "csrc", f), "rb").read())
    for f in TEACHER_HOST:          # (the compiler flags and the block -> operand-format map decide the arithmetic too)
        h.update(open(os.path.join(ROOT, "cosa_amd", f), "rb").read())
    return h.hexdigest()[:16]


TEACHER_CSRC = ("gemm_kernels.hip", "attn_kernels.hip", "split_kernels.hip", "vit_kernels.hip", "label_kernels.hip", "c4.hpp", "c8.hpp",
                "op16.hpp", "common.hpp", "kernels.hpp")
TEACHER_HOST = ("build.py", "models/vit.py")
# COSA_ACCURACY_DATASET=COCO: the same checks with BASELINE configs[3] / [4]'s class count (81 CAM planes per set instead of 21; 80 labels in the
# synthetic batch) -- on record in its own file, the VOC record keeps its format
DATASET = os.environ.get("COSA_ACCURACY_DATASET", "VOC12")
NCLS = {"VOC12": 21, "COCO": 81}[DATASET]
RECORD = os.path.join(ROOT, "gpurun_out", "r06_accuracy_teacher.txt" if DATASET == "VOC12" else "r06_accuracy_teacher_coco.txt")

# ---- the conformance criterion: PRE-REGISTERED -- these constants were committed before any round-6 draw was looked at --------
# BASELINE.json: "within 1e-3 relative on fp32 CAMs ... mask IoU vs. CPU reference >= 0.999".  Per active (image, class) plane of a CAM set:
#   (a) the LITERAL bar: max |delta| of the min-max normalised plane (what multi_scale_camseg returns, seg_helper.py:264-270) <= CAM_BAR.
#   (b) a plane over that bar may take the EXEMPTION only if its conditioning (rawmax / peak from the ORACLE: the magnitude of the class logits
#       over what the normalisation divides by) is > COND_MAX, and then it must keep BOTH own-scale err (|delta| x peak / rawmax) <= CAM_BAR AND
#       |HIP - float64 oracle| <= CAM_BAR + FP64_FACTOR x |fp32 oracle - float64 oracle| on that plane: the exemption is bounded by how far the
#       reference's own fp32 arithmetic is from exact arithmetic there, not by nothing.  Anything else is a FAILED plane.
#   (c) masks: label agreement >= AGREE_BAR on every draw; mask IoU the way the reference scores it (utils/evaluation.py:17-35): ONE confusion matrix
#       pooled over all draws of a (mode, crop, CAM set), per-class IoU = diag / (row + col - diag) over the classes the oracle's masks contain
#       (255 = "ignore" counted as a category of its own), their mean >= MIOU_BAR.  The per-draw mIoU stays on record as a diagnostic.
# A mode is CONFORMING at a crop only if no plane failed, (c) holds, and the record has >= MIN_DRAWS draws of it (bench.py: `tolerance_met`).
CAM_BAR = 1e-3
COND_MAX = 50.0
FP64_FACTOR = 4.0
AGREE_BAR = 0.999
MIOU_BAR = 0.999
MIN_DRAWS = 64
_POOL = {}          # (mode, S, set name) -> {class: [tp, fp, fn]} over the draws of this pytest process (test_pooled_mask_iou_of_this_run)


def _record(lines):
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    new = not os.path.exists(RECORD)
    with open(RECORD, "a") as f:          # (written before the asserts: a failing mode is on record too)
        if new:
            f.write(f"# csrc_sha256_16={teacher_csrc_hash()}  (tests/test_precision_gpu.py: fused HIP teacher vs the fp32 CPU oracle)\n")
        f.write("\n".join(lines) + "\n")


def _oracle_pass(S, seed=3, b=2):
    """fp32 CPU oracle (oracle/torch_oracle.py + cosa_oracle.c) on ViT-B weights drawn with `seed` and the synthetic batch of seed + 2
    (seed 3 / batch 5 are the rounds 2-3 pair), cached per (S, seed, b)"""
    if b != 2:
        return _oracle_pass_b(S, seed, b)
    if (S, seed) not in _ORACLE:
        for k in [k for k in _ORACLE if k != (S, seed)]:          # (cases are ordered by oracle pass: one live entry is enough)
            _ORACLE.pop(k)
        from oracle import torch_oracle as to, c_oracle
        from cosa_amd.models import build_model
        from cosa_amd.train_step import default_args, synthetic_batch
        torch.manual_seed(seed)
        net = build_model(default_args(DATASET, crop_size=S, compute_dtype=torch.float32))
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        wimg, _, lab, box = synthetic_batch(2, S, NCLS - 1, torch.device("cpu"), seed=seed + 2, dataset=DATASET)
        m = to.OracleViT(num_classes=NCLS, aux_layer=default_args(DATASET).aux_layer)          # (run_voc.sh: -4; COCO keeps args.py's -3)
        m.load_named(sd)
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        with torch.no_grad():
            cam, cam_aux, _, sc_cam, sc_aux = to.multi_scale_camseg(m, wimg, [1.0, 0.5, 1.5], return_scale=True)
        bx = np.asarray(box.numpy(), np.int32)
        masks = [c_oracle.cam2mask(None, bx, c.numpy(), lab.numpy(), 0.7, 0.25, 2, par=None) for c in (cam, cam_aux)]
        _ORACLE[(S, seed)] = (sd, wimg, lab, box, cam, cam_aux, masks, (sc_cam, sc_aux))
    return _ORACLE[(S, seed)]


def _oracle_pass_b(S, seed, b):
    """the same for another batch size (the bench's own b = 16), two images at a time on the CPU; cached under (S, seed, b)"""
    if (S, seed, b) not in _ORACLE:
        _ORACLE.clear()
        from oracle import torch_oracle as to, c_oracle
        from cosa_amd.models import build_model
        from cosa_amd.train_step import default_args, synthetic_batch
        torch.manual_seed(seed)
        net = build_model(default_args(DATASET, crop_size=S, compute_dtype=torch.float32))
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        wimg, _, lab, box = synthetic_batch(b, S, NCLS - 1, torch.device("cpu"), seed=seed + 2, dataset=DATASET)
        m = to.OracleViT(num_classes=NCLS, aux_layer=default_args(DATASET).aux_layer)          # (run_voc.sh: -4; COCO keeps args.py's -3)
        m.load_named(sd)
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        cams, auxs, scs = [], [], []
        with torch.no_grad():
            for i in range(0, b, 2):
                c, ca, _, sc, sa = to.multi_scale_camseg(m, wimg[i:i + 2], [1.0, 0.5, 1.5], return_scale=True)
                cams.append(c)
                auxs.append(ca)
                scs.append((sc, sa))
        cam, cam_aux = torch.cat(cams), torch.cat(auxs)
        cat2 = lambda k, j: torch.cat([t[k][j] for t in scs])
        bx = np.asarray(box.numpy(), np.int32)
        masks = [c_oracle.cam2mask(None, bx, c.numpy(), lab.numpy(), 0.7, 0.25, 2, par=None) for c in (cam, cam_aux)]
        _ORACLE[(S, seed, b)] = (sd, wimg, lab, box, cam, cam_aux, masks, ((cat2(0, 0), cat2(0, 1)), (cat2(1, 0), cat2(1, 1))))
    return _ORACLE[(S, seed, b)]


_ORACLE64 = {}


def _oracle_fp64_planes(S, seed, b, images):
    """float64 leg of the oracle for the exempted planes: the SAME OracleViT weights / inputs in float64 (oracle/torch_oracle.py), for the image
    pairs that hold `images`; returns {image: (cam64 [C, S, S], cam_aux64)} normalised planes.  Cached per (S, seed, b, pair)."""
    from oracle import torch_oracle as to
    from cosa_amd.train_step import default_args
    sd, wimg = _oracle_pass(S, seed, b)[:2]
    out = {}
    for i in sorted({(im // 2) * 2 for im in images}):
        key = (S, seed, b, i)
        if key not in _ORACLE64:
            for k in [k for k in _ORACLE64 if k[:3] != (S, seed, b)]:
                _ORACLE64.pop(k)
            m = to.OracleViT(num_classes=NCLS, aux_layer=default_args(DATASET).aux_layer)
            m.load_named(sd)
            m = m.double()
            with torch.no_grad():
                cam, cam_aux, _ = to.multi_scale_camseg(m, wimg[i:i + 2].double(), [1.0, 0.5, 1.5])
            _ORACLE64[key] = (cam, cam_aux)
        cam, cam_aux = _ORACLE64[key]
        for j in range(cam.shape[0]):
            out[i + j] = (cam[j], cam_aux[j])
    return out


def _confusion(pred, true, n):
    """per class (0 .. n-1 and 255): [tp, fp, fn] of the HIP mask against the oracle's"""
    out = {}
    for c in list(range(n)) + [255]:
        P, T = pred == c, true == c
        tp, fp, fn = int((P & T).sum()), int((P & ~T).sum()), int((~P & T).sum())
        if tp or fp or fn:
            out[c] = [tp, fp, fn]
    return out


def pooled_miou(conf):
    """conf: {class: [tp, fp, fn]} summed over draws -> (mean IoU over the classes present in the oracle's masks, min IoU, classes)"""
    ious = [v[0] / (v[0] + v[1] + v[2]) for v in conf.values() if v[0] + v[2] > 0]
    return (float(np.mean(ious)), float(np.min(ious)), len(ious)) if ious else (1.0, 1.0, 0)


# mode -> (max normalised-CAM relative error, min label agreement, min mask IoU); the conforming modes carry the north-star bars
NORTH_STAR = (1e-3, 0.999, 0.999)      # BASELINE.json north_star: 1e-3 relative on fp32 CAMs, mask IoU >= 0.999
ON_RECORD = (2e-3, 0.999, 0.998)       # maps measured for the margin table: faster, but over the bar on at least one seed / crop
TEACHER_BARS = {
    "bf16": (3e-2, 0.99, 0.97),
    "fp16": (4e-3, 0.9990, 0.995),
    "bf16x3": NORTH_STAR,
    "fp16x3": NORTH_STAR,          # round 6: the three-term path with fp16 halves (11 + 11 significant bits), same cost as bf16x3
    "fp16c8": NORTH_STAR,          # fp16 + two e5m2 correction terms (round 3): 3 failed planes on the round-6 record (conditioning 43 / 66 / 2199)
    "fp16c4": NORTH_STAR,          # round 4: FP4 (e2m1, MX blocks) correction terms in qkv / fc1 / fc2 of every block
    "fp16c8-x2": NORTH_STAR,       # round 5's default (fp16c8 with blocks 0-1 on bf16x3 operands): literal worst 7.3e-4 on seeds 100-139, but 3 failed
    #                                planes on the round-6 record (224^2 seed 112: 1.75e-3 at conditioning 43; b = 16 seed 8; seed 210): not the default any more
    "fp16c4-12m9": ON_RECORD,      # round 4's default: holds the bars on the seven seeds it was chosen on, fails 12 of 40 held-out draws
    "fp16c8-9": ON_RECORD,         # round 3's benchmarked mode (last three blocks plain fp16): 1.06e-3 at 640^2 on one seed
    "fp16c4-10": ON_RECORD, "fp16c4-9": ON_RECORD,
    "fp16c4-8": ON_RECORD,         # last four blocks plain fp16: 2.0x inside the bar on seeds 3 / 11 / 29, mask IoU 0.9988 on seed 17
    "fp16c4-12m8": ON_RECORD,      # MLP halves plain from block 8: the auxiliary CAM (block 8's output) reaches 9.4e-4 on seed 23
    "fp16c4-10q": ON_RECORD,       # plain-fp16 qkv projections in the corrected blocks (-0.2 ms per block, 1.7x the error)
}
# the modes a bench line may carry as its headline are checked on SEVEN independent weight / batch draws (three were not enough: the maps with
# plain-fp16 attention in the last blocks pass seeds 3 / 11 / 29 with a 2x margin and fail seed 17); bench.py reports the WORST of these lines
# (profiles/r04_accuracy_teacher.txt, copied from gpurun_out/ after the GPU run) and derives `tolerance_met` from them
CONFORMING_SEEDS = (3, 11, 29, 5, 17, 23, 41)
_MULTI = {448: ("fp16x3", "fp16c8-x2", "fp16c8"), 224: ("fp16x3", "fp16c8-x2", "fp16c8")}      # (round 4 also ran fp16c4 / fp16c8-9 / fp16c4-8 on all seven: profiles/r04_accuracy_teacher.txt)
_HISTORIC = ("fp16c4-10", "fp16c4-9", "fp16c4-12m8", "fp16c4-10q")          # round 4's margin table: on record in profiles/r04_accuracy_teacher.txt, not re-run
ALL_SEEDS = os.environ.get("COSA_ACCURACY_ALL_SEEDS", "0") == "1"        # the evidence run (tools/accuracy_evidence.sh): all seven seeds, four at 640^2
SUITE_SEEDS = (11, 29, 5, 17, 23, 41) if ALL_SEEDS else (11, 29, 17, 41)          # the suite re-asserts four of round 4's seven seeds besides seed 3 (suite time); the record holds all seven
# suite time (the driver's GPU run): the on-record modes of earlier rounds (fp16c4 family, fp16c8-9) and uniform fp16c8 on the extra seeds run in the
# evidence run only (COSA_ACCURACY_ALL_SEEDS=1: tools/accuracy_evidence.sh 1); the suite keeps every operand family once (seed 3) and the default +
# round 5's default on the four extra seeds
_EVIDENCE_ONLY = () if ALL_SEEDS else ("fp16c4", "fp16c4-12m9", "fp16c8-9", "fp16c4-8")
_MULTI_SUITE = {S: tuple(m for m in ms if ALL_SEEDS or m != "fp16c8") for S, ms in _MULTI.items()}
_CASES = [(m, 3, S) for m in TEACHER_BARS if m not in _HISTORIC + _EVIDENCE_ONLY for S in (224, 448)] + \
    [(m, sd_, S) for S, ms in _MULTI_SUITE.items() for m in ms for sd_ in SUITE_SEEDS]


@pytest.mark.parametrize("mode,seed,S", sorted(_CASES, key=lambda c: (c[2], c[1])))       # (grouped by oracle pass)
def test_fused_teacher_vs_fp32_cpu_oracle(mode, seed, S):
    _check_teacher(mode, seed, S)


# at 640^2 (3601 tokens at scale 1.5) the maps with plain-fp16 blocks lose most: fp16c4-8 9.8e-4, fp16c8-9 1.06e-3 on seed 11 -- on record only;
# the trainer's "auto" default takes fp16c8 above 448^2
MODES_640 = (("fp16c4-12m9", False), ("fp16c8", False), ("fp16c8-x2", False), ("fp16x3", True))


@pytest.mark.parametrize("seed", (3, 11, 17, 29) if ALL_SEEDS else (11,))          # (the committed record also holds seeds 3, 17 and 29: the evidence run)
def test_fused_teacher_vs_fp32_cpu_oracle_640(seed):
    """the crop of BASELINE configs[4] (COCO, 640^2: 1601 / 401 / 3601 tokens per image and scale), so that a bench line at --crop 640 has its
    accuracy evidence too"""
    try:
        for mode, must_conform in MODES_640:
            _check_teacher(mode, seed, 640, bars=NORTH_STAR if must_conform else ON_RECORD)
    finally:
        _ORACLE.pop((640, seed), None)          # (7-MB CAM sets and their inputs: not needed again)


def _check_teacher(mode, seed, S, bars=None, b=2):
    """One draw: the fused HIP teacher in operand mode `mode` against the fp32 CPU oracle, by the pre-registered criterion at the top of this
    file (CAM_BAR / COND_MAX / FP64_FACTOR / AGREE_BAR / MIOU_BAR).  Per CAM set (main, aux) one record line: the worst literal figure
    (normalised-CAM rel err), the worst own-scale err and conditioning, label agreement, the draw's own mIoU (diagnostic), then
    `| planes N literal-ok L exempt E fail F` with one bracket per plane that is over the literal bar (conditioning, literal, own-scale, and --
    where the exemption applies -- |HIP - fp64| against its bound CAM_BAR + FP64_FACTOR x |fp32 - fp64|), then the confusion counts
    `conf class:tp/fp/fn,...` of the HIP mask against the oracle's that the pooled IoU is formed from.
    Modes carrying NORTH_STAR are asserted by that criterion; the on-record / historic modes by their own looser per-draw bars (gross-error gates)."""
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    from cosa_amd.utils import seg_helper
    sd, wimg, lab, box, cam_o, cam_aux_o, masks_o, scales_o = _oracle_pass(S, seed, b)
    args = default_args(DATASET, crop_size=S)
    net = build_model(args).cuda().eval()
    net.load_state_dict(sd)
    net.set_nograd_precision(mode)
    assert net.encoder.use_fused(wimg.cuda()) or torch.is_grad_enabled()
    with torch.no_grad():
        cam, cam_aux, _ = seg_helper.multi_scale_camseg(net, wimg.cuda(), args.pseudo_scales)
        masks = [seg_helper.cam2mask(wimg.cuda(), box, c * lab.cuda()[:, :, None, None], lab.cuda(), 0.7, 0.25).cpu().numpy() for c in (cam, cam_aux)]
    act = lab.bool()
    bar_rel, bar_agree, bar_iou = bars or TEACHER_BARS[mode]
    strict = (bar_rel, bar_agree, bar_iou) == NORTH_STAR
    lines, failed = [], []
    sets = (("cam", cam.cpu(), cam_o, masks[0], masks_o[0], scales_o[0], 0), ("cam_aux", cam_aux.cpu(), cam_aux_o, masks[1], masks_o[1], scales_o[1], 1))
    # planes over the literal bar whose conditioning admits the exemption: they need the float64 leg (strict modes only: the others are on record
    # with the literal / own-scale figures alone)
    over = {}
    for name, g, o, _, _, (peak, rawmax), _ in sets:
        d = (g - o).abs().amax(dim=(2, 3))
        lit = d / o.abs().amax(dim=(2, 3)).clamp_min(1e-6)
        cond = rawmax / peak
        over[name] = [(int(i), int(c)) for i, c in zip(*torch.nonzero(act & (lit > CAM_BAR), as_tuple=True))]
    need64 = sorted({i for name in over for (i, c) in over[name]
                     if float((scales_o[0] if name == "cam" else scales_o[1])[1][i, c] / (scales_o[0] if name == "cam" else scales_o[1])[0][i, c]) > COND_MAX}) if strict else []
    o64 = _oracle_fp64_planes(S, seed, b, need64) if need64 else {}
    for name, g, o, mg, mo, (peak, rawmax), k in sets:
        d = (g - o).abs().amax(dim=(2, 3))
        lit = d / o.abs().amax(dim=(2, 3)).clamp_min(1e-6)
        own = d * peak / rawmax.clamp_min(1e-30)
        cond = rawmax / peak
        n_planes, n_ok, n_ex, n_fail, notes = int(act.sum()), int((act & (lit <= CAM_BAR)).sum()), 0, 0, []
        for (i, c) in over[name]:
            cd, lt, ow = float(cond[i, c]), float(lit[i, c]), float(own[i, c])
            if cd <= COND_MAX or not strict:
                n_fail += 1
                notes.append(f"[img {i} cls {c}: cond {cd:.1f} lit {lt:.3e} own {ow:.3e} FAIL]")
                continue
            p64 = o64[i][k][c]
            e_ref = float((o[i, c].double() - p64).abs().max())
            e_hip = float((g[i, c].double() - p64).abs().max())
            bound = CAM_BAR + FP64_FACTOR * e_ref
            ok = ow <= CAM_BAR and e_hip <= bound
            n_ex, n_fail = n_ex + ok, n_fail + (not ok)
            notes.append(f"[img {i} cls {c}: cond {cd:.1f} lit {lt:.3e} own {ow:.3e} fp64 {e_hip:.3e} bound {bound:.3e} {'ok' if ok else 'FAIL'}]")
        agree, iou = float(np.mean(mg == mo)), _miou(mg, mo, NCLS)
        conf = _confusion(mg, mo, NCLS)
        pool = _POOL.setdefault((mode, S, name), {})
        for c, v in conf.items():
            pool[c] = [x + y for x, y in zip(pool.get(c, [0, 0, 0]), v)]
        lines.append(f"teacher {mode:8s} S={S} b={b} seed={seed:<2d} {name:8s}: normalised-CAM rel err {float(lit[act].max()):.3e}  label agreement {agree:.5f}  mask mIoU {iou:.5f}"
                     f"  own-scale err {float(own[act].max()):.3e}  worst conditioning {float(cond[act].max()):.1f}"
                     f"  | planes {n_planes} literal-ok {n_ok} exempt {n_ex} fail {n_fail} {' '.join(notes)}"
                     f" | conf {','.join(f'{c}:{v[0]}/{v[1]}/{v[2]}' for c, v in sorted(conf.items()))}")
        # the pre-registered criterion is ENFORCED for the trainer's default and for fp16x3; the other NORTH_STAR modes go through it too and
        # are on record with their failed planes (bench.conformance reads them), but in the suite they answer for the per-draw gross gates
        # only: on a plane of conditioning > 100 the literal figure of a 14-bit mode moves by 1e-3 with the last bit of the inputs
        enforce = strict and mode in (_auto_mode(S), "fp16x3")
        if enforce and (n_fail or agree < AGREE_BAR):
            failed.append(lines[-1])
        if not enforce:
            g_rel, g_agree, g_iou = (1e-3, 0.999, 0.998) if strict else (bar_rel, bar_agree, bar_iou)
            if not (float(own[act].max()) <= g_rel and agree >= g_agree and iou >= g_iou):
                failed.append(lines[-1])
    _record(lines)          # (written before the asserts: a failing mode is on record too)
    assert not failed, "\n".join(failed)


# ---- the wide sweep behind the headline: >= 32 further weight / batch draws at 448^2 and ONE batch of the bench's own size
# (b = 16).  In the suite by default: SWEEP_DEFAULT draws (suite time); `COSA_ACCURACY_SWEEP_SEEDS=40 pytest -k sweep` writes the committed record
# (profiles/r06_accuracy_teacher.txt, tools/accuracy_evidence.sh).  Seeds 100-139 are disjoint from CONFORMING_SEEDS, on which round 4's block maps
# were chosen -- they are what showed that fp16c4-12m9 and uniform fp16c8 do not hold, and the set the round-5 default (fp16c8-x2) was picked on;
# seeds 200-231 (COSA_ACCURACY_SWEEP_BASE=200) were drawn after that choice; seeds 300-323 are round 6's, drawn after the criterion at the top
# of this file was fixed and before fp16x3 existed: held-out for both.
SWEEP_DEFAULT = 1
SWEEP_SEEDS = int(os.environ.get("COSA_ACCURACY_SWEEP_SEEDS", str(SWEEP_DEFAULT)))
SWEEP_BASE = int(os.environ.get("COSA_ACCURACY_SWEEP_BASE", "100"))        # 100-139: the draws the round-5 map was chosen on; 200-231: drawn after the choice
SWEEP_S = int(os.environ.get("COSA_ACCURACY_SWEEP_S", "448"))               # crop size of the sweep (the record also holds sweeps at 224 and 640)
SWEEP_MODES = tuple(os.environ.get("COSA_ACCURACY_SWEEP_MODES", "fp16x3,fp16c8-x2").split(","))


def _auto_mode(S):
    from cosa_amd.train_step import resolve_teacher_precision
    return resolve_teacher_precision("auto", S)


@pytest.mark.parametrize("seed", [SWEEP_BASE + i for i in range(SWEEP_SEEDS)])
def test_headline_modes_on_held_out_seeds_sweep(seed):
    """every mode of the sweep is put on record; the trainer's `auto` choice at 448^2 must hold the north-star bars on every draw"""
    auto = _auto_mode(SWEEP_S)
    err = None
    for mode in dict.fromkeys(SWEEP_MODES + (auto,)):
        try:
            _check_teacher(mode, seed, SWEEP_S, bars=NORTH_STAR)
        except AssertionError as e:
            if mode == auto:
                err = e
    if err is not None:
        raise err


@pytest.mark.skipif(os.environ.get("COSA_ACCURACY_B16", "1") != "1", reason="COSA_ACCURACY_B16=0: the b = 16 draw (50 s of CPU oracle) switched off")
@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("COSA_ACCURACY_B16_SEEDS", "7").split(",")])
def test_headline_mode_on_the_bench_batch_b16(seed):
    """BASELINE configs[1] itself: b = 16 x 448^2 through the fused teacher (M = 87 904 token rows per pass, the launch shapes of the bench)
    against the fp32 CPU oracle run two images at a time"""
    modes = [m for m in os.environ.get("COSA_ACCURACY_B16_MODES", "").split(",") if m] or [_auto_mode(448)]      # (the evidence run puts more modes on record)
    err = None
    try:
        for mode in modes:
            try:
                _check_teacher(mode, seed, 448, bars=NORTH_STAR, b=16)
            except AssertionError as e:
                if mode == _auto_mode(448):
                    err = e
    finally:
        _ORACLE.clear()
    if err is not None:
        raise err


def test_pooled_mask_iou_of_this_run():
    """criterion (c) on whatever this pytest process drew (runs after the per-draw tests above in file order; the committed record's pooled
    figure over ALL draws is checked by bench.conformance / tests/test_boundary.py): per (mode, crop, CAM set) the confusion matrix pooled over
    the draws, per-class IoU from it the way utils/evaluation.py:17-35 forms it, mean >= MIOU_BAR for the modes that carry NORTH_STAR"""
    if not _POOL:
        pytest.skip("no teacher draw ran in this process")
    bad = []
    for (mode, S, name), conf in sorted(_POOL.items()):
        miou, lo, n = pooled_miou(conf)
        if mode in (_auto_mode(S), "bf16x3") and miou < MIOU_BAR:          # (the other modes are on record; a sweep swallows their failures)
            bad.append(f"{mode} S={S} {name}: pooled mIoU {miou:.5f} (min class IoU {lo:.5f}, {n} classes)")
    assert not bad, bad


@pytest.mark.parametrize("seed", (3, 17))
def test_teacher_masks_through_par_vs_fp32_cpu_oracle(seed):
    """--usepar: the label maps after PAR refinement (PAR.py:64-91: ten affinity-propagation steps over the CAMs) from the fused HIP teacher in
    the default mode against the oracle's (fp32 CAMs, oracle/cosa_oracle.c PAR) on the same denormalised images: the refinement must not
    amplify the teacher's operand rounding past the mask bar (IoU >= 0.999, agreement >= 0.999), S = 448"""
    from cosa_amd.models import build_model
    from cosa_amd.models.PAR import PAR
    from cosa_amd.train_step import default_args, resolve_teacher_precision
    from cosa_amd.utils import seg_helper, torch_helper
    from oracle import c_oracle
    S, DIL = 448, [1, 2, 4, 8, 12, 24]
    sd, wimg, lab, box, cam_o, cam_aux_o, _, _ = _oracle_pass(S, seed)
    args = default_args("VOC12", crop_size=S)
    net = build_model(args).cuda().eval()
    net.load_state_dict(sd)
    mode = resolve_teacher_precision("auto", S, usepar=True)          # (fp16c8: fp16c4-12m9 gives mask IoU 0.99899 on seed 17 after PAR)
    net.set_nograd_precision(mode)
    img = torch_helper.denormalize_img(wimg.cuda())
    bx = np.asarray(box.numpy(), np.int32)
    with torch.no_grad():
        cam, cam_aux, _ = seg_helper.multi_scale_camseg(net, wimg.cuda(), args.pseudo_scales)
        par = PAR(num_iter=10, dilations=DIL).cuda()
        got = [seg_helper.cam2mask(img, box, c * lab.cuda()[:, :, None, None], lab.cuda(), 0.7, 0.25, refine_model=par).cpu().numpy() for c in (cam, cam_aux)]
    ref = [c_oracle.cam2mask(img.cpu().numpy(), bx, c.numpy(), lab.numpy(), 0.7, 0.25, 2, par=(DIL, 10)) for c in (cam_o, cam_aux_o)]
    lines = []
    for name, mg, mo in (("cam", got[0], ref[0]), ("cam_aux", got[1], ref[1])):
        agree, iou = float(np.mean(mg == mo)), _miou(mg, mo)
        lines.append(f"teacher+PAR {mode:8s} S={S} b=2 seed={seed:<2d} {name:8s}: label agreement {agree:.5f}  mask mIoU {iou:.5f}")
    with open(os.path.join(ROOT, "gpurun_out", "r06_accuracy_teacher_par.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")
    for ln in lines:
        agree, iou = (float(ln.split(k)[1].split()[0]) for k in ("label agreement", "mask mIoU"))
        assert agree >= 0.999 and iou >= 0.999, ln


def test_c8_operand_buffers_are_keyed_on_the_token_geometry():
    """the token-shaped patch operand of the fp16c8 / fp16c4 passes keeps its class-token rows zero "for good", so its cache must
    be keyed on the (images, tokens) split and not on the row count M alone: two batches with the SAME M and another split (2 images of 4 x 4
    patches, then 1 image of 3 x 11: 34 token rows each) must give what a fresh model gives, and the cache stays bounded"""
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    torch.manual_seed(0)
    a = default_args("VOC12", crop_size=64)
    net = build_model(a).cuda().eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    xa, xb = torch.randn(2, 3, 64, 64, device="cuda"), torch.randn(1, 3, 48, 176, device="cuda")
    for mode in ("fp16c4", "fp16c8"):
        net.set_nograd_precision(mode)
        with torch.no_grad():
            net.forward_multi([xa])
            got = [t.clone() for t in net.forward_multi([xb])[0] if t is not None]
        fresh = build_model(a).cuda().eval()
        fresh.load_state_dict(sd)
        fresh.set_nograd_precision(mode)
        with torch.no_grad():
            want = [t for t in fresh.forward_multi([xb])[0] if t is not None]
        for g, w in zip(got, want):
            assert torch.equal(g, w), mode
    with torch.no_grad():
        for wd in range(1, 14):                      # 13 more geometries: the cache holds at most _C8_BUFS_MAX of them
            net.forward_multi([torch.randn(1, 3, 16, 16 * wd, device="cuda")])
    assert len(net.encoder._c8_bufs) <= net.encoder._C8_BUFS_MAX
