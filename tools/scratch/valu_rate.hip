// per-SIMD issue rate of the softmax's VALU instructions (round 5): cycles per wave-instruction measured with
// s_memtime around a block of independent instructions, 1 / 2 / 4 waves per SIMD.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/scratch/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long *cyc, float *sink, const float *in)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = in[(threadIdx.x + i) & 255];
    h2 hp = {(_Float16)a[0], (_Float16)a[1]};
    float acc = a[2];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; it++) {
        if (OP == 0) { REP64(asm volatile("v_exp_f32 %0, %1" : "=v"(a[0]) : "v"(a[1]));) }            // independent (same source)
        if (OP == 1) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[0]) : "v"(a[1]), "v"(a[2]), "v"(a[3]));) }
        if (OP == 2) { REP64(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a[1]), "v"(a[2]));) }
        if (OP == 3) { REP64(asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(hp), "v"(hp));) }   // (accumulating: a dependent chain, as in the kernel's four chains / 4)
        if (OP == 4) { REP64(asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[0]) : "v"(a[1]), "v"(a[2]), "v"(a[3]));) }
        if (OP == 5) { REP64(asm volatile("v_exp_f16 %0, %1" : "=v"(hp) : "v"(hp));) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a[0] + (float)hp[0] + acc;
}
template <int OP>
void run(const char *name, unsigned long long *cyc, float *sink, float *in)
{
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {          // 256 threads = 4 waves = one per SIMD
        const int blocks = 256 * wg_per_cu;
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, cyc, sink, in);
        hipDeviceSynchronize();
        static unsigned long long h[8192];
        hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
        // __builtin_readcyclecounter = s_memtime: shader cycles
        const double per = s / (blocks * 4) / 4096.0;
        printf("%-18s %d wave(s)/SIMD: %6.2f cycles per instruction seen by a wave -> one instruction per %5.2f cycles per SIMD\n", name, wg_per_cu, per, per / wg_per_cu);
    }
}
int main()
{
    float *in, *sink;
    unsigned long long *cyc;
    hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    hipMalloc(&sink, 256 * 2048 * 4 * 4);
    hipMalloc(&cyc, 8 * 8192 * 4);
    run<1>("v_fma_f32", cyc, sink, in);
    run<0>("v_exp_f32", cyc, sink, in);
    run<5>("v_exp_f16", cyc, sink, in);
    run<2>("v_cvt_pk_f16_f32", cyc, sink, in);
    run<3>("v_dot2c_f32_f16", cyc, sink, in);
    run<4>("v_max3_f32", cyc, sink, in);
    return 0;
}
