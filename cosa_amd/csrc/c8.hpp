// c8.hpp -- "fp16c8" operands: an fp16 value plus 8-bit correction terms (parity-grade no-grad passes at 2x instead of 3x MFMA work).
//
// A value v is carried as hi = fp16(v) (11 significant bits) and two 8-bit floats (e5m2, OCP bf8):
//     hi8 = e5m2(hi)                   a 3-bit copy of the hi part
//     lo8 = e5m2((v - hi) * 2^11)      what the fp16 rounding dropped, scaled into hi's range (|v - hi| <= 2^-11 |v|)
// and a product of two such values is   x_hi w_hi  +  2^-11 (x_lo8 w_hi8 + x_hi8 w_lo8):
// the first term is the fp16 MFMA the 1x path runs, the two correction terms need only ~3 bits each (they are 2^-11 of the result) and
// run on gfx950's block-scaled 8-bit MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, K = 128 per instruction, twice the fp16 rate) with the
// 2^-11 applied by the instruction's E8M0 scale operand.  e5m2 has fp16's exponent range, so no per-tensor statistics / scale
// management exist anywhere: every conversion is a clamp and a round.  Measured against the fp32 CPU oracle (tools/sim_precision_map.py,
// tests/test_precision_gpu.py) the scheme sits between fp16 (1.5e-3 on the normalised CAMs at 448^2) and bf16x3 (2e-5).
//
// Row layout of an operand of logical width K (bytes; the row stride equals the bf16x3 layout's, (2K + 64) * 2):
//     [ hi fp16 (2K) | lo8 (K) | hi8 (K) | aug fp16 (128) ]        aug = (1, 1, 0, ...) for activations, (bias_hi, bias_lo, 0, ...) for weights
// In 128-byte column tiles (Kp = K / 64, Kh = Kp / 2): hi = tiles [0, Kp), lo8 = [Kp, Kp + Kh), hi8 = [Kp + Kh, 2 Kp), aug = tile 2 Kp.
#pragma once

namespace cosa {

constexpr float kC8LoScale = 2048.0f;          // 2^11
constexpr float kC8Max = 57344.0f;             // largest finite e5m2

// Saturation (round 6: ONE v_med3_f32 on the value instead of a min / max pair on each of its two 8-bit terms).  With |v| <= 57344 (which is an
// fp16 value) hi = fp16(v) is finite and <= 57344 in magnitude, and |v - hi| <= half an fp16 ulp = 2^-11 2^15 at most, so lo = (v - hi) 2^11 <=
// 2^15 < 57344: neither term can leave e5m2's range and the conversions need no clamp of their own.  For every |v| <= 57344 the bytes are the
// ones the clamped form produced (its clamps were identities there); beyond, v saturates as a whole instead of going through fp16 infinity.
__device__ __forceinline__ float c8_sat(float v) { return __builtin_amdgcn_fmed3f(v, -kC8Max, kC8Max); }

// four floats (each within +-57344) -> four e5m2 bytes (little end first), round to nearest even
__device__ __forceinline__ unsigned c8_pack4(float a, float b, float c, float d)
{
    int r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
    return (unsigned)r;
}

// v[4] -> hi (fp16 x4), packed lo8, packed hi8
__device__ __forceinline__ void c8_split4(const float (&v)[4], _Float16 (&hi)[4], unsigned &lo8, unsigned &hi8)
{
    float h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float vs = c8_sat(v[j]);
        hi[j] = (_Float16)vs;
        h[j] = (float)hi[j];
        l[j] = (vs - h[j]) * kC8LoScale;
    }
    lo8 = c8_pack4(l[0], l[1], l[2], l[3]);
    hi8 = c8_pack4(h[0], h[1], h[2], h[3]);
}

}  // namespace cosa
