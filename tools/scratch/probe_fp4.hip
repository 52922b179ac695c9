// probe_fp4.hip -- what gfx950's fp4 instructions do, before the fp16c4 operand format is built on them (round 4):
//   (1) v_cvt_scalef32_pk_fp4_f32: rounding (nearest even on the e2m1 grid?), saturation, the meaning of the scale operand, nibble order
//   (2) v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 operands: which K indices a lane's 32 nibbles stand for, and that a lane's scale byte
//       (selected by op_sel) applies to exactly those 32 values
// build: hipcc -shared -fPIC --offload-arch=gfx950 -O3 tools/scratch/probe_fp4.hip -o tools/scratch/libprobe_fp4.so ; run: probe_fp4.py
#include <hip/hip_runtime.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void cvt_kernel(const float *f, const float *scale, unsigned *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, f[2 * i], f[2 * i + 1], scale[i], 0);
    out[i] = r;
}

// A: [16][128] nibble codes as i32x4 per lane (lane = row + 16 g holds k in [32 g, 32 g + 32), nibble p of the 16 bytes = k = 32 g + p),
// B likewise (lane = col + 16 g).  sa / sb: one int per lane, byte `sel` holds the lane's E8M0 scale.
template <int SEL>
__global__ void mfma_kernel(const i32x4 *a, const i32x4 *b, const int *sa, const int *sb, f32x4 *d)
{
    const int l = threadIdx.x;
    const i32x4 av = a[l], bv = b[l];
    const i32x8 a8 = {av.x, av.y, av.z, av.w, 0, 0, 0, 0}, b8 = {bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc, 4, 4, SEL, sa[l], SEL, sb[l]);
    d[l] = acc;
}

extern "C" void probe_cvt(const float *f, const float *scale, unsigned *out, int n, void *stream)
{
    hipLaunchKernelGGL(cvt_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, f, scale, out, n);
}
extern "C" void probe_mfma(const void *a, const void *b, const int *sa, const int *sb, void *d, int sel, void *stream)
{
    if (sel == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const i32x4 *)a, (const i32x4 *)b, sa, sb, (f32x4 *)d);
    else if (sel == 1) hipLaunchKernelGGL(mfma_kernel<1>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const i32x4 *)a, (const i32x4 *)b, sa, sb, (f32x4 *)d);
    else if (sel == 2) hipLaunchKernelGGL(mfma_kernel<2>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const i32x4 *)a, (const i32x4 *)b, sa, sb, (f32x4 *)d);
    else hipLaunchKernelGGL(mfma_kernel<3>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const i32x4 *)a, (const i32x4 *)b, sa, sb, (f32x4 *)d);
}
