"""probe of gfx950's fp4 conversion and fp4 block-scaled MFMA (see probe_fp4.hip); prints what it finds, asserts nothing"""
import ctypes
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libprobe_fp4.so"))
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
GRID = np.array([0, 0.5, 1, 1.5, 2, 3, 4, 6.0])


def e2m1_rne(a):
    a = np.minimum(np.abs(a), 6.0)
    r = np.where(a < 2, np.round(a * 2) / 2, np.where(a < 4, np.round(a), np.round(a / 2) * 2))      # np.round = half to even
    return r


def code_of(v):          # value -> 4-bit code (sign << 3 | index in GRID)
    idx = np.array([int(np.where(GRID == abs(x))[0][0]) for x in v])
    return idx | ((np.signbit(v)).astype(np.int64) << 3)


# ---- (1) conversion ------------------------------------------------------------------------------------------------
vals = np.array([0, 0.24, 0.25, 0.26, 0.5, 0.74, 0.75, 0.76, 1.0, 1.24, 1.25, 1.26, 1.5, 1.74, 1.75, 1.76, 2, 2.49, 2.5, 2.51, 3, 3.49, 3.5, 3.51,
                 4, 4.99, 5, 5.01, 6, 6.99, 7, 8, 100, -0.3, -1.25, -2.5, -5, -7, 1e-30, float("inf")], np.float32)
vals = np.concatenate([vals, np.zeros(len(vals) % 2, np.float32)])
for scale in (1.0, 4.0, 0.5, 3.0):
    f = torch.from_numpy(vals * (scale if scale != 3.0 else 2.0)).to(dev)
    n = f.numel() // 2
    s = torch.full((n,), scale, device=dev)
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    L.probe_cvt(p(f), p(s), p(out), n, st)
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.int64)
    lo, hi = o & 0xF, (o >> 4) & 0xF
    got = np.stack([lo, hi], 1).reshape(-1)
    mag = GRID[got & 7] * np.where(got & 8, -1, 1)
    exp = e2m1_rne(vals) * np.sign(vals)
    print(f"scale {scale}: input/scale_pow2 -> fp4 value (expected RNE)")
    print("  " + "  ".join(f"{v:g}->{m:g}({e:g})" for v, m, e in zip(vals, mag, exp)))
    print("  mismatches vs RNE e2m1 of x/scale:", int((mag != exp).sum()), " first element in LOW nibble:", bool((got[0::2] == lo).all()))

# ---- (2) MFMA ------------------------------------------------------------------------------------------------------
rng = np.random.default_rng(0)
for sel in range(4):
    A = rng.integers(0, 16, (16, 128))          # codes
    B = rng.integers(0, 16, (16, 128))          # [col][k]
    ea = rng.integers(120, 134, (16, 4))        # E8M0 exponents per (row, 32-block)
    eb = rng.integers(120, 134, (16, 4))
    dec = lambda c: GRID[c & 7] * np.where(c & 8, -1.0, 1.0)
    Af = dec(A) * np.repeat(2.0 ** (ea - 127.0), 32, axis=1)
    Bf = dec(B) * np.repeat(2.0 ** (eb - 127.0), 32, axis=1)
    ref = Af @ Bf.T                               # [row][col]
    def pack(C):                                  # lane = r + 16 g: 16 bytes, nibble p (low first) = k = 32 g + p
        out = np.zeros((64, 16), np.uint8)
        for g in range(4):
            for r in range(16):
                c = C[r, 32 * g:32 * g + 32]
                out[r + 16 * g] = (c[0::2] | (c[1::2] << 4)).astype(np.uint8)
        return out
    def scales(E):
        out = np.zeros(64, np.int64)
        junk = rng.integers(0, 255, 64)
        for g in range(4):
            for r in range(16):
                v = 0
                for byte in range(4):
                    v |= int(E[r, g] if byte == sel else junk[r + 16 * g]) << (8 * byte)
                out[r + 16 * g] = v
        return out.astype(np.uint32).view(np.int32) if False else np.array(out, dtype=np.uint32).view(np.int32)
    a_t = torch.from_numpy(pack(A)).to(dev)
    b_t = torch.from_numpy(pack(B)).to(dev)
    sa_t = torch.from_numpy(scales(ea)).to(dev)
    sb_t = torch.from_numpy(scales(eb)).to(dev)
    d = torch.zeros((64, 4), device=dev)
    L.probe_mfma(p(a_t), p(b_t), p(sa_t), p(sb_t), p(d), sel, st)
    torch.cuda.synchronize()
    D = d.cpu().numpy()                           # lane l: col = l & 15, rows 4 (l >> 4) + reg
    got = np.zeros((16, 16))
    for l in range(64):
        for r in range(4):
            got[4 * (l >> 4) + r, l & 15] = D[l, r]
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"mfma fp4 op_sel {sel}: max rel err vs natural-k reference {err:.3e}  (transposed: {np.abs(got.T - ref).max() / np.abs(ref).max():.3e})")
