// op16.hpp -- the 16-bit MFMA operand type of a translation unit.
//
// gemm_kernels.hip and attn_kernels.hip are compiled twice: with bf16 operands (default; the training path of the student and the
// pure-bf16 teacher) and, with -DCOSA_OP_F16=1, with IEEE fp16 operands (entry points suffixed _f16): the no-grad teacher /
// evaluation passes may run on fp16 operands -- same MFMA rate, 3 more mantissa bits -- which is what brings the pseudo-label
// maps within the stated tolerance of the fp32 reference.  Accumulation is fp32 either way.
#pragma once

#if COSA_OP_F16
typedef _Float16 op16;
#define COSA_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define COSA_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#else
typedef __bf16 op16;
#define COSA_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define COSA_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif
// half of the type's coarsest relative spacing (one unit in the last place of a value just above a power of two, halved): x + |x| * this
// rounds to an op16 value >= x
#if COSA_OP_F16
constexpr float kOp16HalfSpacing = 0.00048828125f;          // 2^-11
#else
constexpr float kOp16HalfSpacing = 0.00390625f;             // 2^-8
#endif
typedef op16 op16x8 __attribute__((ext_vector_type(8)));
typedef op16 op16x4 __attribute__((ext_vector_type(4)));
typedef op16 op16x2 __attribute__((ext_vector_type(2)));

// the two packed 16-bit values of a dword as floats
__device__ __forceinline__ float op16_lo(unsigned u)
{
#if COSA_OP_F16
    return (float)__builtin_bit_cast(op16x2, u)[0];
#else
    return __builtin_bit_cast(float, u << 16);
#endif
}
__device__ __forceinline__ float op16_hi(unsigned u)
{
#if COSA_OP_F16
    return (float)__builtin_bit_cast(op16x2, u)[1];
#else
    return __builtin_bit_cast(float, u & 0xffff0000u);
#endif
}
