"""GPU tests of the fp16c4 operand format (round 4; csrc/c4.hpp): fp16 + FP4 (e2m1) correction terms in MX blocks on the block-scaled MFMA.

The format is restated here in numpy / torch (fields, block order, scale rule, round-to-nearest-even on the e2m1 grid, the scale tensor's
panel layout) and every producer is compared with it FIELD BY FIELD, bit for bit; the GEMM is compared with the float64 value of its own
terms (decoded operands), so a wrong lane -> k map, a wrong scale byte or a wrong op_sel shows as an O(1) error, not as noise.
Reference for the arithmetic it replaces: models/vit/vit.py:96-137 (fp32 nn.Linear / GELU / LayerNorm)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRID = np.array([0, 0.5, 1, 1.5, 2, 3, 4, 6.0])


def _block_exp(amax):
    """smallest e with amax <= 6 * 2^e (c4.hpp: c4_block_exp), amax float32 array"""
    u = amax.astype(np.float32).view(np.uint32).astype(np.int64)
    e = (u >> 23) - 127 - 2 + ((u & 0x7fffff) > 0x400000)
    return np.clip(e, -120, 120)


def _e2m1_codes(v, scale):
    """round-to-nearest-even of v / scale on the e2m1 grid -> 4-bit codes (sign << 3 | grid index)"""
    a = np.abs(v.astype(np.float64) / scale)
    a = np.minimum(a, 6.0)
    r = np.where(a < 2, np.round(a * 2) / 2, np.where(a < 4, np.round(a), np.round(a / 2) * 2))
    idx = np.searchsorted(GRID, r)
    assert np.array_equal(GRID[idx], r)
    return (idx | (np.signbit(v).astype(np.int64) << 3)).astype(np.uint8)


def _decode(codes):
    return GRID[codes & 7] * np.where(codes & 8, -1.0, 1.0)


def ref_c4(src, weight):
    """src fp32 [R, K] -> dict(hi fp16 [R, K], blocks uint8 [R, K] (16 bytes per 16 features), exp int [R, K / 16])"""
    R, K = src.shape
    hi = src.astype(np.float16)
    h = hi.astype(np.float32)
    lo = ((src - h) * np.float32(2048.0)).astype(np.float32)
    hb, lb = h.reshape(R, K // 16, 16), lo.reshape(R, K // 16, 16)
    amax = np.maximum(np.abs(hb).max(2), np.abs(lb).max(2))
    e = _block_exp(amax)
    sc = (2.0 ** e)[..., None]
    ch, cl = _e2m1_codes(hb, sc), _e2m1_codes(lb, sc)
    pack = lambda c: (c[..., 0::2] | (c[..., 1::2] << 4)).astype(np.uint8)                # value 0 in the low nibble of byte 0
    first, second = (pack(ch), pack(cl)) if weight else (pack(cl), pack(ch))
    blocks = np.concatenate([first, second], axis=2).reshape(R, K)
    return dict(hi=hi, blocks=blocks, exp=e, h=h, lo4=_decode(cl) * sc, hi4=_decode(ch) * sc)


def scale_offsets(R, K, weight):
    """byte offset of the scale of (row r, block b of the row) in the scale tensor (c4.hpp: c4_scale_off_x / _w)"""
    Kq = K // 128
    r = np.arange(R)[:, None]
    b = np.arange(K // 16)[None, :]
    q, j = b // 8, b % 8
    P, rr = r >> 8, r & 255
    fq, ks, frow = j & 3, j >> 2, rr & 15
    if weight:
        qa, wr, i = rr >> 7, (rr >> 6) & 1, (rr >> 4) & 3
        inner = ((((wr * 16 + frow) * 4 + fq) * 2 + qa) * 4 + i) * 2 + ks
    else:
        qb, wc, jj = rr >> 7, (rr >> 5) & 3, (rr >> 4) & 1
        inner = ((((wc * 16 + frow) * 4 + fq) * 2 + qb) * 2 + jj) * 2 + ks
    return ((P * Kq + q) << 11) + inner


def split_fields(rows, R, K):
    """c4 rows tensor (fp16 [R, 2K + 64]) -> (hi fp16 [R, K], blocks uint8 [R, K], aug fp16 [R, 64])"""
    raw = rows.view(torch.uint8).reshape(R, 4 * K + 128).cpu().numpy()
    hi = raw[:, :2 * K].copy().view(np.float16)
    return hi, raw[:, 2 * K:3 * K].copy(), raw[:, 4 * K:].copy().view(np.float16)


def decode_operand(rows, scales, R, K, weight):
    """-> float64 (hi, lo' * scale, hi4 * scale) as the MFMA sees them (the weights' 2^-11 taken out of the scale again)"""
    hi, blocks, _ = split_fields(rows, R, K)
    sb = scales.cpu().numpy()[scale_offsets(R, K, weight)].astype(np.int64)
    e = sb - 127 + (11 if weight else 0)
    blk = blocks.reshape(R, K // 16, 16)
    nib = np.stack([blk & 15, blk >> 4], axis=3).reshape(R, K // 16, 32)                   # 32 values per block, low nibble first
    first, second = _decode(nib[..., :16]), _decode(nib[..., 16:])
    sc = (2.0 ** e)[..., None]
    hi4, lo4 = (first, second) if weight else (second, first)
    return hi.astype(np.float64), (lo4 * sc).reshape(R, K), (hi4 * sc).reshape(R, K), sb


def _check_fields(rows, scales, src, weight, bias=None, ones=False):
    R, K = src.shape
    ref = ref_c4(src, weight)
    hi, blocks, aug = split_fields(rows, R, K)
    assert np.array_equal(hi.view(np.uint16), ref["hi"].view(np.uint16))
    got_sb = scales.cpu().numpy()[scale_offsets(R, K, weight)].astype(np.int64)
    want_sb = np.clip(ref["exp"] + 127 + (-11 if weight else 0), 0, 254)
    assert np.array_equal(got_sb, want_sb)
    bad = np.argwhere(blocks != ref["blocks"])
    assert bad.size == 0, (bad[:5], blocks[tuple(bad[0])], ref["blocks"][tuple(bad[0])])
    want_aug = np.zeros((R, 64), np.float16)
    if ones:
        want_aug[:, :2] = 1
    elif bias is not None:
        want_aug[:, 0] = bias.astype(np.float16)
        want_aug[:, 1] = (bias - want_aug[:, 0].astype(np.float32)).astype(np.float16)
    assert np.array_equal(aug.view(np.uint16), want_aug.view(np.uint16))


@pytest.mark.parametrize("R,K,weight", [(300, 768, False), (256, 3072, True), (700, 256, True), (5, 768, False)])
def test_c4_rows_fields_bit_identical_to_the_format(R, K, weight):
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(R + K)
    src = torch.randn(R, K, generator=g) * torch.exp(2 * torch.randn(R, 1, generator=g))          # rows of very different scale
    src[0, :40] = 0
    src[min(3, R - 1), 5] = 1e-30
    bias = torch.randn(R, generator=g)
    rows, scales = nn_ops.c4_rows(src.cuda(), bias=bias.cuda() if weight else None, ones=not weight, weight=weight)
    _check_fields(rows, scales, src.numpy(), weight, bias=bias.numpy() if weight else None, ones=not weight)


def test_c4_rows_batched_equals_single_launches():
    import ctypes
    from cosa_amd import nn_ops, _C
    g = torch.Generator().manual_seed(1)
    mats = [(torch.randn(256, 768, generator=g).cuda(), torch.randn(256, generator=g).cuda()),
            (torch.randn(768, 3072, generator=g).cuda() * 0.02, torch.randn(768, generator=g).cuda())]
    rec_dt = np.dtype([("src", "u8"), ("bias", "u8"), ("dst", "u8"), ("sc", "u8"), ("rows", "i4"), ("K", "i4"), ("row0", "i4"), ("qrows", "i4")])
    assert rec_dt.itemsize == _C.lib().cosa_c4_record_bytes()
    rec, outs, row0 = np.zeros(len(mats), rec_dt), [], 0
    qrows = [0, 256]                 # second matrix: the first 256 rows (and bias entries) leave multiplied by 64^-0.5 log2(e) (folded attention scale)
    for j, (w, b) in enumerate(mats):
        o = torch.zeros((w.shape[0], nn_ops.split_ld(w.shape[1])), device="cuda", dtype=torch.float16)
        sc = nn_ops.c4_scales(w.shape[0], w.shape[1], "cuda")
        outs.append((o, sc))
        rec[j] = (w.data_ptr(), b.data_ptr(), o.data_ptr(), sc.data_ptr(), w.shape[0], w.shape[1], row0, qrows[j])
        row0 += w.shape[0]
    d_rec = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    _C.check(_C.lib().cosa_c4_rows_batched(_C.ptr(d_rec), len(mats), row0, _C.stream_ptr()), "cosa_c4_rows_batched")
    c = torch.tensor(0.125, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
    for (w, b), (o, sc), qr in zip(mats, outs, qrows):
        w, b = w.clone(), b.clone()
        w[:qr] *= c
        b[:qr] *= c
        o1, sc1 = nn_ops.c4_rows(w, bias=b, weight=True)
        assert torch.equal(o.view(torch.int16), o1.view(torch.int16)) and torch.equal(sc, sc1)


@pytest.mark.parametrize("rows", [1000, 4099])
def test_layernorm_c4_fields(rows):
    """LayerNorm -> c4 rows: the fp32 result equals cosa_layernorm_c8's, and the c4 fields are the format's fields of that fp32 result"""
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, 768, generator=g) * 3 + 1).cuda()
    gam, bet = (1 + 0.3 * torch.randn(768, generator=g)).cuda(), (0.2 * torch.randn(768, generator=g)).cuda()
    out = torch.zeros((rows, nn_ops.split_ld(768)), device="cuda", dtype=torch.float16)
    sc = nn_ops.c4_scales(rows, 768, "cuda")
    _, y32 = nn_ops.layernorm_c4(x, gam, bet, 1e-6, out=out, scales=sc, want_f32=True)
    o8 = torch.zeros_like(out)
    _, y32_8 = nn_ops.layernorm_c8(x, gam, bet, 1e-6, out=o8, want_f32=True)
    assert torch.equal(y32, y32_8)
    _check_fields(out, sc, y32.cpu().numpy(), False, ones=True)


def _gelu64(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K,epi", [(1000, 768, 768, 0), (5000, 2304, 768, 0), (4608, 768, 3072, 2), (2900, 3072, 768, 1), (87904 // 8, 768, 768, 2)])
def test_gemm_c4_vs_fp64_of_its_own_terms(M, N, K, epi):
    """cosa_gemm_f16c4 == x_hi w_hi + bias + 2^-11 (x_lo4 w_hi4 + x_hi4 w_lo4) evaluated in float64 from the DECODED operands (so the MFMA's
    lane -> k map, the scale ring, the op_sel of every scale byte and the tile order are all on the line), all three epilogues; and against
    float64 of the fp32 inputs: ~2^-13 relative (the scheme's accuracy).  GELU epilogue: the c4 rows it writes are well-formed (scale bytes and
    hi nibbles recomputed from the stored fp16 hi parts bit for bit) and decode to gelu(y) within 2^-12."""
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(M + N + K + epi)
    x = torch.randn(M, K, generator=g) * torch.exp(0.7 * torch.randn(M, 1, generator=g))
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.3
    res = torch.randn(M, N, generator=g) if epi == 2 else None
    xs, xsc = nn_ops.c4_rows(x.cuda(), ones=True)
    ws, wsc = nn_ops.c4_rows(w.cuda(), bias=b.cuda(), weight=True)
    out = nn_ops.gemm_c4(xs, xsc, ws, wsc, M, N, K, epi, residual=res.cuda() if res is not None else None)
    torch.cuda.synchronize()
    xh, xl, xh4, _ = decode_operand(xs, xsc, M, K, False)
    wh, wl, wh4, _ = decode_operand(ws, wsc, N, K, True)
    b16 = b.numpy().astype(np.float16)
    bias_terms = b16.astype(np.float64) + (b.numpy() - b16.astype(np.float32)).astype(np.float16).astype(np.float64)
    own = xh @ wh.T + bias_terms + (xl @ wh4.T + xh4 @ wl.T) / 2048.0
    exact = x.double().numpy() @ w.double().numpy().T + b.double().numpy()
    scale = np.abs(exact).max()
    if epi == 2:
        got = out.cpu().double().numpy() - res.double().numpy()
        assert np.abs(got - own).max() <= 3e-6 * scale, np.abs(got - own).max() / scale
        assert np.abs(got - exact).max() <= 2.5e-4 * scale, np.abs(got - exact).max() / scale
    elif epi == 0:
        got = out.cpu().double().numpy()
        assert np.abs(got - own).max() <= 6e-4 * scale                                   # fp16 output rounding
        assert np.abs(got - exact).max() <= 8e-4 * scale
    else:
        rows, ysc = out
        yh, yl, yh4, sb = decode_operand(rows, ysc, M, N, False)
        ref = _gelu64(exact)
        val = yh + yl / 2048.0
        gs = np.abs(ref).max()
        assert np.abs(yh - ref).max() <= 1.2e-3 * gs                                      # the fp16 part alone: fp16 rounding + operand accuracy
        # the lo' nibbles carry real correction: hi + lo' / 2^11 is closer to the kernel's own fp32 value than hi alone for large entries
        own_g = _gelu64(own)
        big = np.abs(own_g) > 0.25 * gs
        assert np.abs(val - own_g)[big].mean() < 0.5 * np.abs(yh - own_g)[big].mean()
        # well-formed blocks: the scale byte and the hi nibbles follow from the stored fp16 hi parts
        hb = yh.astype(np.float32).reshape(M, N // 16, 16)
        e = _block_exp(np.abs(hb).max(2))
        assert np.array_equal(sb, np.clip(e + 127, 0, 254))
        want_h4 = _decode(_e2m1_codes(hb, (2.0 ** e)[..., None])) * (2.0 ** e)[..., None]
        assert np.array_equal(yh4.reshape(M, N // 16, 16), want_h4)


def test_gemm_c4_is_bit_identical_run_to_run_and_more_accurate_than_fp16():
    from cosa_amd import nn_ops
    g = torch.Generator().manual_seed(7)
    M, N, K = 6000, 768, 768
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.04, torch.randn(N, generator=g)
    res = torch.zeros(M, N).cuda()
    xs, xsc = nn_ops.c4_rows(x.cuda(), ones=True)
    ws, wsc = nn_ops.c4_rows(w.cuda(), bias=b.cuda(), weight=True)
    a = nn_ops.gemm_c4(xs, xsc, ws, wsc, M, N, K, 2, residual=res)
    c = nn_ops.gemm_c4(xs, xsc, ws, wsc, M, N, K, 2, residual=res)
    assert torch.equal(a, c)
    exact = x.double() @ w.double().t() + b.double()
    y16 = nn_ops.gemm_bf16(x.cuda().half(), w.cuda().half(), b.cuda().half(), 2, residual=res)
    e4, e16 = (a.cpu().double() - exact).abs().max().item(), (y16.cpu().double() - exact).abs().max().item()
    assert e4 < 0.3 * e16, (e4, e16)


@pytest.mark.parametrize("B,N,row0", [(2, 197, 0), (3, 785, 512), (1, 300, 77)])
def test_attention_c4_output_rows(B, N, row0):
    """cosa_attn_fwd_f16c4 against cosa_attn_fwd_f16c8 on the same fp16 qkv: the fp16 hi parts are the same bits (same attention arithmetic),
    the blocks are well-formed (scale bytes -- at the rows' place in the whole operand's scale tensor, offset row0 -- and hi nibbles follow
    from the stored hi parts bit for bit), and the lo' nibbles agree with the c8 rows' e5m2 lo bytes within their quantisation steps"""
    from cosa_amd import nn_ops
    H, D = 12, 768
    g = torch.Generator().manual_seed(B * N)
    qkv = (torch.randn(B, N, 3 * D, generator=g) * 0.7).cuda().half()
    o8 = torch.zeros((B * N, nn_ops.split_ld(D)), device="cuda", dtype=torch.float16)
    nn_ops.attn_fwd_c8(qkv, B, N, H, o8)
    R = row0 + B * N + 5
    full = torch.zeros((R, nn_ops.split_ld(D)), device="cuda", dtype=torch.float16)
    sc = nn_ops.c4_scales(R, D, "cuda")
    nn_ops.attn_fwd_c4(qkv, B, N, H, full[row0:row0 + B * N], sc, row0)
    torch.cuda.synchronize()
    hi4_, blocks, aug = split_fields(full, R, D)
    raw8 = o8.view(torch.uint8).reshape(B * N, 4 * D + 128).cpu().numpy()
    hi8rows = raw8[:, :2 * D].copy().view(np.float16)
    sl = slice(row0, row0 + B * N)
    assert np.array_equal(hi4_[sl].view(np.uint16), hi8rows.view(np.uint16))
    want_aug = np.zeros((B * N, 64), np.float16)
    want_aug[:, :2] = 1
    assert np.array_equal(aug[sl].view(np.uint16), want_aug.view(np.uint16))
    yh, yl, yh4, sb = decode_operand(full, sc, R, D, False)
    hb = yh[sl].astype(np.float32).reshape(B * N, D // 16, 16)
    e = _block_exp(np.abs(hb).max(2))
    assert np.array_equal(sb[sl], np.clip(e + 127, 0, 254))
    scale = (2.0 ** e)[..., None]
    assert np.array_equal(yh4[sl].reshape(B * N, D // 16, 16), _decode(_e2m1_codes(hb, scale)) * scale)
    lo8 = torch.from_numpy(raw8[:, 2 * D:3 * D].copy()).view(torch.float8_e5m2).float().numpy()          # (v - hi) * 2^11 in e5m2
    lo4 = yl[sl].reshape(B * N, D // 16, 16)
    step = np.maximum(scale * 2.0, np.abs(lo8.reshape(B * N, D // 16, 16)) * 0.26)                           # e2m1 step <= 2 scale; e5m2 step 25 %
    assert (np.abs(lo4 - lo8.reshape(B * N, D // 16, 16)) <= step + 1e-12).all()
    assert (sc.cpu().numpy()[scale_offsets(R, D, False)][:row0] == 0).all()                                  # rows outside the launch untouched


@pytest.mark.parametrize("fmt,K", [("c4", 768), ("c4", 3072), ("c8", 768)])
def test_tail_jobs_of_the_n768_launches_are_bit_identical_to_the_persistent_kernel(fmt, K):
    """The teacher's N = 768 projections are 1032 jobs = 4 x 256 + 8.  For fp16c8 operands (the output projection) the persistent kernel stops
    after four full rounds and the eight leftover 256 x 256 jobs run as 128 x 128 quarters on the two-stage kernel (round 4); fp16c4 launches
    run all 1032 jobs on the persistent kernel.  Either way a row's result may not depend on the launch geometry: the same rows through a
    launch without leftovers (fewer rows) are the same bits."""
    from cosa_amd import nn_ops
    M, N, M2 = 87904, 768, 83808           # 1032 jobs (tail of 8) vs 984 jobs (no tail)
    g = torch.Generator().manual_seed(K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.04).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    if fmt == "c4":
        xs, xsc = nn_ops.c4_rows(x, ones=True)
        ws, wsc = nn_ops.c4_rows(w, bias=b, weight=True)
        full = nn_ops.gemm_c4(xs, xsc, ws, wsc, M, N, K, 2, residual=res)
        part = nn_ops.gemm_c4(xs[:M2], xsc, ws, wsc, M2, N, K, 2, residual=res[:M2].contiguous())
    else:
        xs, ws = nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b)
        full = nn_ops.gemm_c8(xs, ws, M, N, K, 2, residual=res)
        part = nn_ops.gemm_c8(xs[:M2], ws, M2, N, K, 2, residual=res[:M2].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(full[:M2], part)
    exact = x[:4096].double() @ w.double().t() + b.double() + res[:4096].double()
    assert float((full[:4096].double() - exact).abs().max()) <= 3e-4 * float(exact.abs().max())
