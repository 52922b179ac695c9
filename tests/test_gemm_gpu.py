"""GPU parity of the projection GEMM kernels (cosa_gemm_bf16 variants, cosa_gemm_wgrad_bf16, cosa_layernorm) against fp32 torch.

bf16 operands, fp32 accumulation: the only differences from the fp32 reference are the bf16 rounding of the output
(2^-9 relative) and the summation order, so the tolerance is 2^-8 of the output scale for bf16 results and 1e-5 for fp32 ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(4099, 768, 768), (8192, 256, 64), (4608, 2304, 768), (5000, 768, 3072), (300, 768, 768), (257, 256, 128), (1, 128, 64),
          (70000, 768, 128), (66000, 256, 192),          # > 256 tiles: the persistent kernel (v6) runs several jobs per workgroup
          (16384, 3072, 768), (18432, 2304, 768)]        # tiles_m % 8 == 0 (the shapes the optional banded tile order applies to)


def _ref(x, w, b, epi, r):
    y = x.float() @ w.float().t() + b.float()
    if epi == 1:
        y = torch.nn.functional.gelu(y)
    if epi == 2:
        y = y + r
    return y


@pytest.mark.parametrize("variant", [0, 1, 6])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_bf16_variants_vs_fp32(variant, epi):
    from cosa_amd import nn_ops, _C
    torch.manual_seed(variant * 10 + epi)
    try:
        for (M, N, K) in SHAPES:
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            r = torch.randn(M, N, device="cuda") if epi == 2 else None
            # canaries after the operands: a tail tile that read past row M would pick these up instead of zeros
            _C.lib().cosa_gemm_set_variant(variant)
            y = nn_ops.gemm_bf16(x, w, b, epi, residual=r)
            ref = _ref(x, w, b, epi, r)
            assert y.dtype == (torch.float32 if epi == 2 else torch.bfloat16)
            tol = (1e-5 if epi == 2 else 2.0 ** -8) * max(ref.abs().max().item(), 1.0)
            err = (y.float() - ref).abs().max().item()
            assert err <= tol, (variant, epi, M, N, K, err, tol)
    finally:
        _C.lib().cosa_gemm_set_variant(0)


@pytest.mark.parametrize("M,N,K", [(12560, 768, 768), (1000, 256, 128), (63, 128, 128), (4097, 2304, 768)])
def test_gemm_wgrad_vs_fp32(M, N, K):
    from cosa_amd import nn_ops
    torch.manual_seed(M)
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dw, db = nn_ops.gemm_wgrad(dy, x, want_bias=True)
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    assert dw.dtype == torch.float32 and dw.shape == (N, K)
    assert (dw - ref_w).abs().max().item() <= 2e-4 * ref_w.abs().max().item() + 1e-3
    assert (db - ref_b).abs().max().item() <= 2e-4 * ref_b.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K", [(12560, 2304, 768), (12560, 768, 3072), (12544, 768, 768), (700, 128, 256)])
def test_gemm_wgrad_is_bit_identical_run_to_run_and_accumulates(M, N, K):
    """round 3: the token splits of the weight gradient meet in a fixed-order reduction of fp32 slabs instead of fp32 atomics -- the same
    bits every run (also a race screen for the three-stage LDS-DMA ring with its counted vmcnt); zero_first = 0 adds to what is there"""
    from cosa_amd import nn_ops, _C
    torch.manual_seed(K)
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    first_w, first_b = (t.clone() for t in nn_ops.gemm_wgrad(dy, x, want_bias=True))
    for _ in range(20):
        w, b = nn_ops.gemm_wgrad(dy, x, want_bias=True)
        assert torch.equal(w, first_w) and torch.equal(b, first_b)
    L = _C.lib()
    ws = _C.workspace(L.cosa_gemm_wgrad_workspace_bytes(M, N, K), dy.device, "wgrad")
    base_w, base_b = torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    acc_w, acc_b = base_w.clone(), base_b.clone()
    _C.check(L.cosa_gemm_wgrad_bf16(_C.ptr(dy), _C.ptr(x), _C.ptr(acc_w), _C.ptr(acc_b), M, N, K, 0, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "wgrad")
    scale = first_w.abs().max().item()
    assert (acc_w - (base_w + first_w)).abs().max().item() <= 1e-5 * scale and (acc_b - (base_b + first_b)).abs().max().item() <= 1e-4 * first_b.abs().max().item()


def test_layernorm_vs_torch():
    from cosa_amd import nn_ops
    torch.manual_seed(5)
    x = torch.randn(3001, 768, device="cuda") * 2 + 0.5
    g = torch.randn(768, device="cuda").bfloat16()
    b = torch.randn(768, device="cuda").bfloat16()
    y16, y32 = nn_ops.layernorm_f32(x, g, b, 1e-6, True, True)
    ref = torch.nn.functional.layer_norm(x, (768,), g.float(), b.float(), 1e-6)
    assert (y32 - ref).abs().max().item() < 2e-5
    assert (y16.float() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()


@pytest.mark.parametrize("variant", [6])
def test_gemm_pipelined_kernels_are_run_to_run_deterministic(variant):
    """race screen for the counted-vmcnt / staggered-barrier schedules: no atomics are involved, so any run-to-run difference of the
    output bits would be an LDS-DMA data race (a buffer read before its DMA landed, or refilled while still being read)"""
    from cosa_amd import nn_ops, _C
    torch.manual_seed(11)
    try:
        _C.lib().cosa_gemm_set_variant(variant)
        for (M, N, K, epi) in [(87904, 768, 768, 2), (20000, 2304, 768, 0), (9000, 3072, 768, 1), (66000, 768, 3072, 2), (4099, 256, 128, 0)]:
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            r = torch.randn(M, N, device="cuda") if epi == 2 else None
            first = nn_ops.gemm_bf16(x, w, b, epi, residual=r).clone()
            ref = x.float() @ w.float().t() + b.float()
            if epi == 1:
                ref = torch.nn.functional.gelu(ref)
            if epi == 2:
                ref = ref + r
            tol = (1e-5 if epi == 2 else 2.0 ** -8) * max(ref.abs().max().item(), 1.0)
            assert (first.float() - ref).abs().max().item() <= tol
            del ref
            for _ in range(25):
                again = nn_ops.gemm_bf16(x, w, b, epi, residual=r)
                assert torch.equal(again, first)
    finally:
        _C.lib().cosa_gemm_set_variant(0)


@pytest.mark.parametrize("B,h,w,Cin,Cout,strided", [(3, 14, 14, 768, 512, True), (2, 28, 28, 512, 512, False), (1, 9, 13, 128, 128, False),
                                                    (5, 5, 7, 256, 128, True)])
def test_dilated_conv_autograd_vs_torch(B, h, w, Cin, Cout, strided):
    """LargeFOV conv (3x3, dilation 5, no bias, ReLU) forward + input gradient + weight gradient on the MFMA kernels (implicit GEMM /
    implicit im2col) against torch's conv2d autograd in fp32 on the same bf16-rounded operands; ragged token counts (M % 64 != 0),
    images smaller than the dilation reach, and the strided token view (encoder tokens without their cls row)."""
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(B * 100 + h)
    full = (torch.randn(B, h * w + 1, Cin, generator=g) * 0.5).to(torch.bfloat16).cuda()
    tok = (full[:, 1:] if strided else full[:, 1:].contiguous()).detach().requires_grad_(True)
    weight = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02).cuda().requires_grad_(True)
    gy = torch.randn(B * h * w, Cout, generator=g).to(torch.bfloat16).cuda()
    y = nn_ops.DilatedConvReluFn.apply(tok, weight, B, h, w, 5)
    y.backward(gy)
    # reference: fp32 conv on the bf16-rounded operands
    x32 = tok.detach().float().reshape(B, h, w, Cin).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w32 = weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = torch.relu(torch.nn.functional.conv2d(x32, w32, padding=5, dilation=5))
    yr.backward(gy.float().reshape(B, h, w, Cout).permute(0, 3, 1, 2))
    yref = yr.detach().permute(0, 2, 3, 1).reshape(B * h * w, Cout)
    assert (y.float() - yref).abs().max() <= 2e-2 * yref.abs().max()
    # the gate uses our bf16 forward's sign; compare gradients where both forwards agree on it (all but measure-zero ties)
    dxr = x32.grad.permute(0, 2, 3, 1).reshape(B, h * w, Cin)
    assert (tok.grad.float() - dxr).abs().max() <= 3e-2 * dxr.abs().max()
    assert (weight.grad - w32.grad).abs().max() <= 3e-2 * w32.grad.abs().max()
    assert weight.grad.dtype == torch.float32 and weight.grad.shape == weight.shape


@pytest.mark.parametrize("dtype,N,K", [(torch.float32, 20, 768), (torch.bfloat16, 21, 512), (torch.float32, 80, 768), (torch.bfloat16, 81, 512),
                                       (torch.bfloat16, 20, 768)])
def test_head_linear_vs_torch_and_batch_invariance(dtype, N, K):
    """narrow heads (CAM / aux-CAM / conv8): fp32-accumulated X W^T on the strided token view (no cls row) against torch in float64;
    a row's result must not depend on the batch around it (bitwise), which library GEMMs do not guarantee"""
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(N + K)
    B, n = 5, 197
    full = torch.randn(B, n + 1, K, generator=g).to(dtype).cuda()
    tok = full[:, 1:]
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    y = nn_ops.head_linear(tok, w, round_bf16=dtype == torch.bfloat16)
    ref = (tok.double().reshape(-1, K) @ w.double().t())
    tol = 2e-6 if dtype == torch.float32 else 1e-2                       # bf16: the result is rounded to bf16 precision
    assert y.shape == (B * n, N) and y.dtype == torch.float32
    assert (y.double() - ref).abs().max() <= tol * ref.abs().max()
    one = nn_ops.head_linear(tok[2:3], w, round_bf16=dtype == torch.bfloat16)
    assert torch.equal(one, y[2 * n:3 * n])
    two = nn_ops.head_linear(tok[3:5].contiguous(), w, round_bf16=dtype == torch.bfloat16)
    assert torch.equal(two, y[3 * n:5 * n])


@pytest.mark.parametrize("M,N,K,act", [(12560, 3072, 768, True), (12560, 768, 3072, False), (300, 2304, 768, False), (130, 3072, 768, True),
                                       (4099, 768, 768, False)])
def test_student_linear_own_kernels_vs_torch_autograd(M, N, K, act):
    """nn.Linear (+GELU) of the student's blocks, forward and backward on own kernels (dual-epilogue GEMM, GELU', the forward kernel on the
    transposed shadow for dX, TN kernel for dW / db) against fp32 torch autograd on the same bf16-rounded operands.
    Tolerances: bf16 outputs 2^-7 of the output scale (two roundings in the GELU chain), fp32 weight gradients 2e-3 relative."""
    from cosa_amd import nn_ops
    torch.manual_seed(M + N)
    lin = torch.nn.Linear(K, N).cuda()
    with torch.no_grad():
        lin.weight.mul_(3.0)
    sh = nn_ops.ShadowSet(lin)
    tsh = nn_ops.TransposedShadows([lin.weight])
    assert torch.equal(tsh.t16[0], lin.weight.detach().bfloat16().t().contiguous())
    x = torch.randn(2, M // 2, K, device="cuda").bfloat16().requires_grad_(True)
    y = nn_ops.linear(x, lin.weight, lin.bias, torch.bfloat16, act=act)
    go = torch.randn_like(y)
    y.backward(go)
    w32, b32 = sh.shadows[0].float().requires_grad_(True), sh.shadows[1].float().requires_grad_(True)
    x32 = x.detach().float().requires_grad_(True)
    hh = torch.nn.functional.linear(x32, w32, b32)
    ref = torch.nn.functional.gelu(hh) if act else hh
    ref.backward(go.float())
    tol = lambda r: 2.0 ** -7 * max(r.abs().max().item(), 1.0)
    assert (y.float() - ref).abs().max().item() <= tol(ref)
    assert (x.grad.float() - x32.grad).abs().max().item() <= 2 * tol(x32.grad)
    assert (lin.weight.grad - w32.grad).abs().max().item() <= 4e-3 * w32.grad.abs().max().item() + 1e-3
    assert (lin.bias.grad - b32.grad).abs().max().item() <= 4e-3 * b32.grad.abs().max().item() + 1e-3


@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_tail_launch_is_interchangeable_with_persistent_jobs(epi):
    """M = 66 000, N = 768: 258 x 3 = 774 jobs = 3 full rounds of 256 + 6 -- the six leftover 256 x 256 jobs run as 128 x 128 quarters on
    the two-stage kernel (launch_v6).  A token's result must not depend on which kernel computed it: rolling the rows of X by half the
    matrix (every row changes tile, most change kernel) rolls the output and nothing else, bit for bit; and the values match fp32."""
    from cosa_amd import nn_ops
    torch.manual_seed(epi)
    M, N, K = 66000, 768, 768
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    sh = 33000 + 128
    y = nn_ops.gemm_bf16(x, w, b, epi, residual=r.clone() if r is not None else None)
    y2 = nn_ops.gemm_bf16(x.roll(sh, 0).contiguous(), w, b, epi, residual=r.roll(sh, 0).contiguous() if r is not None else None)
    assert torch.equal(y2, y.roll(sh, 0))
    ref = _ref(x, w, b, epi, r)
    tol = (1e-5 if epi == 2 else 2.0 ** -8) * max(ref.abs().max().item(), 1.0)
    assert (y.float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(12560, 768, 768), (4099, 768, 3072), (70001, 2304, 768)])
def test_gemm_192_wide_tile_is_bitwise_the_256_wide_tile(epi, M, N, K):
    """variant 9 = the persistent kernel on 256 x 192 jobs (three 16-feature fragments per wave and W half; the student's N = 768
    projections quantise to 200 such jobs instead of 150 of 256 x 256).  Same MFMA chain per output element (bias first, k ascending),
    so nothing may differ from variant 6, not one bit -- one job, many jobs per workgroup (70 001 x 2304: 3288 jobs), ragged M."""
    from cosa_amd import nn_ops, _C
    torch.manual_seed(M + epi)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    try:
        _C.lib().cosa_gemm_set_variant(6)
        y6 = nn_ops.gemm_bf16(x, w, b, epi, residual=r)
        _C.lib().cosa_gemm_set_variant(9)
        # canary-filled output: a store outside the job's window or a missed one shows up
        out = torch.full((M + 8, N), 7.0, device="cuda", dtype=y6.dtype)
        y9 = nn_ops.gemm_bf16(x, w, b, epi, residual=r, out=out[:M])
    finally:
        _C.lib().cosa_gemm_set_variant(0)
    assert torch.equal(y9, y6)
    assert torch.all(out[M:] == 7.0)
    ref = _ref(x, w, b, epi, r)
    tol = (1e-5 if epi == 2 else 2.0 ** -8) * max(ref.abs().max().item(), 1.0)
    assert (y9.float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K", [(12544, 20, 768), (12544, 21, 512), (16, 20, 768), (3001, 80, 768), (777, 81, 512)])
def test_narrow_linear_forward_backward_vs_fp64_autograd(M, N, K):
    """the training path of the narrow heads (CAM / aux-CAM / classification heads, conv8; VOC 20 | 21 and COCO 80 | 81 rows): forward,
    dX = dY W and dW = dY^T X on the exact-fp32 MFMA kernels against float64 autograd on the same bf16-rounded operands.  The only
    roundings left are the bf16 store of dX and fp32 summation order."""
    from cosa_amd import nn_ops
    torch.manual_seed(N + K)
    x = torch.randn(M, K, device="cuda").bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, 1, 1, device="cuda") * K ** -0.5).requires_grad_(True)
    dy = torch.randn(M, N, device="cuda")
    y = nn_ops.narrow_linear(x, w)
    assert y is not None and y.dtype == torch.float32 and y.shape == (M, N)
    y.backward(dy)
    x64 = x.detach().double().requires_grad_(True)
    w64 = w.detach().bfloat16().double().reshape(N, K).requires_grad_(True)
    y64 = x64 @ w64.t()
    y64.backward(dy.double())
    assert (y.double() - y64).abs().max().item() <= 1e-5 * max(1.0, y64.abs().max().item())
    assert x.grad.dtype == torch.bfloat16
    assert (x.grad.double() - x64.grad).abs().max().item() <= 2.0 ** -8 * x64.grad.abs().max().item()
    assert w.grad.shape == w.shape
    assert (w.grad.double().reshape(N, K) - w64.grad).abs().max().item() <= 2e-5 * w64.grad.abs().max().item()
    # deterministic: a second backward gives the same bits (fixed summation tree, no atomics)
    g1 = w.grad.clone()
    w.grad = None
    x.grad = None
    nn_ops.narrow_linear(x, w).backward(dy)
    assert torch.equal(w.grad, g1)


def test_gemm_wgrad_batched_vs_fp32_and_single_launches():
    """the batched weight-gradient launch (one persistent kernel over the tiles of many linears, no split-K) against fp32 references, against
    the single-launch kernel (same products, different summation tree: fp32 rounding only) and run to run (bit-identical)"""
    from cosa_amd import nn_ops
    torch.manual_seed(11)
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (128, 256), (384, 128)]
    for M in (12560, 197, 64):
        pairs = []
        for (N, K) in shapes * (2 if M == 12560 else 1):
            pairs.append((torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, K, device="cuda").bfloat16(), True))
        outs = [(w.clone(), b.clone()) for (w, b) in nn_ops.gemm_wgrad_batched(pairs)]
        for (dy, x, _), (dw, db) in zip(pairs, outs):
            ref_w, ref_b = dy.float().t() @ x.float(), dy.float().sum(0)
            assert (dw - ref_w).abs().max().item() <= 2e-4 * ref_w.abs().max().item() + 1e-3
            assert (db - ref_b).abs().max().item() <= 2e-4 * ref_b.abs().max().item() + 1e-3
            sw, sb = nn_ops.gemm_wgrad(dy, x, want_bias=True)
            assert (dw - sw).abs().max().item() <= 2e-5 * ref_w.abs().max().item() + 1e-4
        for _ in range(3):
            again = nn_ops.gemm_wgrad_batched(pairs)
            assert all(torch.equal(a[0], o[0]) and torch.equal(a[1], o[1]) for a, o in zip(again, outs))
    # more items than one launch's table holds (48): the entry point cuts the list into several launches
    many = [(torch.randn(200, 128, device="cuda").bfloat16(), torch.randn(200, 128 * (1 + i % 2), device="cuda").bfloat16(), i % 3 == 0) for i in range(101)]
    for (dy, x, wb), (dw, db) in zip(many, nn_ops.gemm_wgrad_batched(many)):
        ref_w = dy.float().t() @ x.float()
        assert (dw - ref_w).abs().max().item() <= 2e-4 * ref_w.abs().max().item() + 1e-3
        assert (db is None) == (not wb)
        if wb:
            assert (db - dy.float().sum(0)).abs().max().item() <= 1e-3


def test_deferred_wgrad_gives_the_same_gradients_as_per_layer_launches():
    """student ViT-B blocks, bf16: the gradients of every parameter with the weight gradients deferred to one batched launch
    (nn_ops.DeferredWgrad, the default) against the per-linear launches (defer_wgrad = False)"""
    from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
    dev = torch.device("cuda", 0)
    args = default_args("VOC12", crop_size=224, batch_size=2, teacher_async=False)
    tr = CoSATrainer(args, dev, seed=0)
    enc = tr.student.encoder
    x = torch.randn(2, 3, 224, 224, device=dev)
    grads = {}
    for defer in (True, False):
        enc.defer_wgrad = defer
        tr.student.zero_grad(set_to_none=True)
        cls, tok, aux = enc.forward_features(x.bfloat16())[:3]
        (tok.float().square().mean() + aux.float().mean() + cls.float().sum() * 1e-3).backward()
        grads[defer] = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    enc.defer_wgrad = True
    assert grads[True].keys() == grads[False].keys() and any("blocks.11.mlp.fc2.weight" in k for k in grads[True])
    for k, g in grads[True].items():
        ref = grads[False][k]
        assert (g - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-7, k
