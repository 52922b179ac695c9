// VALU rate probe: separate mul + add chains, scalar f32 vs packed f32 (v_pk_mul_f32 / v_pk_add_f32), many waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void probe(float *out, const float *in, int iters)
{
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = in[(tid + i) & 1023];
    if (PK) {
        f32x2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, m[4];
        for (int i = 0; i < 4; i++) m[i] = (f32x2){in[(tid + 9 + i) & 1023], in[(tid + 17 + i) & 1023]};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    f32x2 p;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(m[c]), "v"((f32x2){a[k], a[k]}));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(p));
                }
        }
        out[tid] = acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1] + acc[2][0] + acc[2][1] + acc[3][0] + acc[3][1];
    } else {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[8];
        for (int i = 0; i < 8; i++) m[i] = in[(tid + 9 + i) & 1023];
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    float p;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(m[c]), "v"(a[k]));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(p));
                }
        }
        out[tid] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    }
}
int main()
{
    float *in, *out;
    hipMalloc(&in, 4096);
    hipMalloc(&out, 256 * 2048 * 4 * 4);
    hipMemset(in, 0, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;         // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    for (int pk = 0; pk < 2; pk++)
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
            else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // element operations: per thread per iteration 64 mul + 64 add
            const double ops = (double)blocks * 256 * iters * 128;
            printf("%s: %.3f ms  %.1f Gop/s-elem  (cycles per wave-instruction pair-of-results: see ratio)\n", pk ? "packed" : "scalar", ms, ops / ms * 1e-6);
        }
    return 0;
}
