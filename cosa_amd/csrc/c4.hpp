// c4.hpp -- "fp16c4" operands (round 4): an fp16 value plus 4-bit correction terms with MX block scales; the parity-grade no-grad passes
// at ~1.5x instead of fp16c8's ~2x MFMA work.
//
// As in c8.hpp a value v is carried as hi = fp16(v) and what the fp16 rounding dropped, lo' = (v - hi) * 2^11 (|lo'| <= |hi|), and a product
// is   x_hi w_hi  +  2^-11 (x_lo' w_hi + x_hi w_lo')   with the first term on the fp16 MFMA.  The two correction terms now run on gfx950's
// block-scaled MFMA with FP4 (e2m1) operands -- v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 4: K = 128 per instruction in the
// cycles of ONE 16x16x32 fp16 MFMA, i.e. 4x the fp16 rate (e5m2: 2x) -- and BOTH terms share one instruction stream: the MX block of 32
// values that a lane feeds to an instruction is
//     activations (B operand):  [ lo'(f0) .. lo'(f15) | hi(f0) .. hi(f15) ]      of 16 consecutive features f0 .. f15
//     weights     (A operand):  [ hi(f0)  .. hi(f15)  | lo'(f0) .. lo'(f15) ]
// so position p of an x block meets position p of the w block of the same 16 features and the instruction's dot product over the 32 positions
// IS  sum_f x_lo' w_hi + x_hi w_lo'.  The 32 values of a block share one power-of-two scale (E8M0 byte): the smallest 2^e with
// amax / 2^e <= 6 (the largest e2m1 magnitude), amax over the block -- since |lo'| <= |hi| that is the scale of the 16 hi values; values
// are rounded to nearest even on the e2m1 grid {0, .5, 1, 1.5, 2, 3, 4, 6} by v_cvt_scalef32_pk_fp4_f32.  The 2^-11 is folded into the
// WEIGHT blocks' scale bytes (exponent - 11).  Accuracy against the fp32 CPU oracle: tools/sim_precision_map.py scheme `h4i`,
// tests/test_precision_gpu.py.
//
// Storage of an operand of R rows and logical width K (K % 256 == 0):
//   rows    [R][4K + 128 bytes] -- the fp16c8 row stride, so buffers and tile indices are shared:
//             [ hi fp16 (2K) | c4 blocks (K: 16 bytes per 16 features) | unused (K) | aug fp16 (128) ]
//           in 128-byte column tiles (Kp = K / 64, Kq = K / 128): hi = tiles [0, Kp), c4 = [Kp, Kp + Kq), aug = tile 2 Kp; c4 tile q holds the
//           blocks of features [128 q, 128 q + 128), block j = its 16-byte slot j.
//   scales  [ceil(R / 256)][Kq][2048 bytes], one E8M0 byte per (row, block), laid out as the persistent GEMM's lanes read them (one
//           ds_read_b64 / ds_read_b128 per lane and tile gives a lane the scales of all its rows for both k-steps; gemm_kernels.hip):
//             activations: row r = 256 P + 128 qb + 32 wc + 16 jj + frow, block j = fq + 4 ks
//                          -> byte ((((wc 16 + frow) 4 + fq) 2 + qb) 2 + jj) 2 + ks   of panel P, tile q
//             weights:     row r = 256 P + 128 qa + 64 wr + 16 i + frow
//                          -> byte ((((wr 16 + frow) 4 + fq) 2 + qa) 4 + i) 2 + ks
#pragma once

namespace cosa {

constexpr float kC4LoScale = 2048.0f;          // 2^11
constexpr int kC4WeightExpBias = -11;          // weights' scale bytes carry the 2^-11 of the correction terms

// exponent e of the block scale 2^e for a block whose largest magnitude is amax: the smallest e with amax <= 6 * 2^e = 1.5 * 2^(e + 2)
__device__ __forceinline__ int c4_block_exp(float amax)
{
    const unsigned u = __builtin_bit_cast(unsigned, amax);
    int e = (int)(u >> 23) - 127 - 2 + ((u & 0x7fffffu) > 0x400000u ? 1 : 0);
    return e < -120 ? -120 : (e > 120 ? 120 : e);         // (zero / denormal blocks: any scale, the values round to 0)
}
__device__ __forceinline__ float c4_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }
__device__ __forceinline__ unsigned c4_scale_byte(int e, int bias)
{
    const int b = e + 127 + bias;
    return (unsigned)(b < 0 ? 0 : (b > 254 ? 254 : b));
}

// eight floats -> eight e2m1 nibbles (value 0 in the low nibble of byte 0), each the nearest (ties to even) grid point of v / 2^e
__device__ __forceinline__ unsigned c4_pack8(const float *v, float scale)
{
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[0], v[1], scale, 0);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[2], v[3], scale, 1);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[4], v[5], scale, 2);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[6], v[7], scale, 3);
    return r;
}

// byte offset of the scale of (row r, c4 tile q, block j of the tile) in an operand's scale tensor
__device__ __forceinline__ size_t c4_scale_off_x(int r, int q, int j, int Kq)
{
    const int P = r >> 8, rr = r & 255, qb = rr >> 7, wc = (rr >> 5) & 3, jj = (rr >> 4) & 1, frow = rr & 15, fq = j & 3, ks = j >> 2;
    return (((size_t)P * Kq + q) << 11) + (size_t)((((((wc * 16 + frow) * 4 + fq) * 2 + qb) * 2 + jj) * 2) + ks);
}
__device__ __forceinline__ size_t c4_scale_off_w(int r, int q, int j, int Kq)
{
    const int P = r >> 8, rr = r & 255, qa = rr >> 7, wr = (rr >> 6) & 1, i = (rr >> 4) & 3, frow = rr & 15, fq = j & 3, ks = j >> 2;
    return (((size_t)P * Kq + q) << 11) + (size_t)((((((wr * 16 + frow) * 4 + fq) * 2 + qa) * 4 + i) * 2) + ks);
}

// 16 consecutive features v[0..15] of one row -> their fp16 hi parts, the 16-byte c4 block and the block's scale exponent.
// weight_order: [hi | lo'] instead of [lo' | hi].
__device__ __forceinline__ void c4_block16(const float (&v)[16], _Float16 (&hi)[16], unsigned (&blk)[4], int &e, bool weight_order)
{
    float h[16], l[16];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        hi[j] = (_Float16)v[j];
        h[j] = (float)hi[j];
        l[j] = (v[j] - h[j]) * kC4LoScale;
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(h[j]), __builtin_fabsf(l[j])));
    }
    e = c4_block_exp(amax);
    const float s = c4_pow2(e);
    const unsigned l0 = c4_pack8(l, s), l1 = c4_pack8(l + 8, s), h0 = c4_pack8(h, s), h1 = c4_pack8(h + 8, s);
    if (weight_order) { blk[0] = h0; blk[1] = h1; blk[2] = l0; blk[3] = l1; }
    else { blk[0] = l0; blk[1] = l1; blk[2] = h0; blk[3] = h1; }
}

}  // namespace cosa
