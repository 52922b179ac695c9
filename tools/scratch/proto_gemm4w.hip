// K-loop rate of a 4-wave workgroup with 128 x 128 output per wave (256 x 256 per workgroup, accumulators 256 registers per lane -> AGPRs),
// LDS-DMA staged 64-KB K-tiles in a 2-stage ring: the structure DESIGN.md names as the one not tried.  Timing only (checksum output).
//   hipcc -O3 --offload-arch=gfx950 proto_gemm4w.hip -o proto_gemm4w && ./proto_gemm4w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int BK = 64, TILE = 256 * 128;       // bytes of one operand tile (256 rows x 64 bf16)
constexpr int STAGE = 2 * TILE;                // A | B
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int PF>
__global__ __launch_bounds__(256) void proto(const bf16 *__restrict__ X, const bf16 *__restrict__ W, float *__restrict__ out, int M, int N, int K,
                                             int jobs_per_wg, int map)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = M / 256, tiles_n = N / 256, nk = K / BK;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    // per-lane source offsets of this wave's 8 + 8 one-KiB pieces (8 rows each): piece q covers rows 8q .. 8q+7
    unsigned vo[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = 8 * (8 * wave + i) + (lane >> 3), ps = lane & 7;
        vo[i] = (unsigned)((row * K + ((ps ^ swz(row)) << 3)) * 2);
    }
    int stage_no = 0;
    for (int jj = 0; jj < jobs_per_wg; jj++) {
        const int job = blockIdx.x + jj * gridDim.x;
        int m0 = (job % tiles_m) * 256, n0 = ((job / tiles_m) % tiles_n) * 256;
        if (map) { const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3; m0 = ((xcd * 4 + (i & 3) + jj * 32) % tiles_m) * 256; n0 = ((i >> 2) % tiles_n) * 256; }   // an XCD's 32 workgroups: 4 m-panels x 8 n-tiles
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)m0 * K), 0, 256 * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)(W + (size_t)n0 * K), 0, 256 * K * 2, 0x00020000);
        auto dma = [&](int kt, int buf) {
            unsigned char *base = smem + buf * STAGE + (8 * wave) * 1024;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void *)(base + i * 1024), 16, vo[i], kt * BK * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void *)(base + TILE + i * 1024), 16, vo[i], kt * BK * 2, 0, 0);
            }
        };
        if (jj == 0) dma(0, stage_no & 1);
        for (int kt = 0; kt < nk; kt++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int buf = stage_no & 1;
            // next tile: of this job, or the first of the next job (the ring never drains).  PF == 2 issues its 16 pieces between the MFMAs below.
            __amdgpu_buffer_rsrc_t nX = rsX, nW = rsW;
            int nkt = kt + 1;
            bool have_next = true;
            if (kt + 1 >= nk) {
                have_next = jj + 1 < jobs_per_wg;
                const int job2 = blockIdx.x + (jj + 1) * gridDim.x;
                int m2 = (job2 % tiles_m) * 256, n2 = ((job2 / tiles_m) % tiles_n) * 256;
                if (map) { const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3; m2 = ((xcd * 4 + (i & 3) + (jj + 1) * 32) % tiles_m) * 256; n2 = ((i >> 2) % tiles_n) * 256; }
                nX = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)m2 * K), 0, 256 * K * 2, 0x00020000);
                nW = __builtin_amdgcn_make_buffer_rsrc((void *)(W + (size_t)n2 * K), 0, 256 * K * 2, 0x00020000);
                nkt = 0;
            }
            unsigned char *nbase = smem + (buf ^ 1) * STAGE + (8 * wave) * 1024;
            auto piece = [&](int p) {                 // p = 0..15: A pieces then B pieces alternate
                const int i = p >> 1;
                if (p & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(nW, (lds_void *)(nbase + TILE + i * 1024), 16, vo[i], nkt * BK * 2, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(nX, (lds_void *)(nbase + i * 1024), 16, vo[i], nkt * BK * 2, 0, 0);
            };
            if (PF != 2 && have_next) {
#pragma unroll
                for (int p = 0; p < 16; p++) piece(p);
            }
            const unsigned char *As = smem + buf * STAGE + (wr * 128) * 128;
            const unsigned char *Bs = smem + buf * STAGE + TILE + (wc * 128) * 128;
            bf16x8 a[2][4], b[2][4];
            auto frags = [&](int ks, int slot) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = 32 * i + r;
                    const int off = row * 128 + (((2 * ks + hh) ^ swz(row)) << 4);
                    a[slot][i] = *reinterpret_cast<const bf16x8 *>(As + off);
                    b[slot][i] = *reinterpret_cast<const bf16x8 *>(Bs + off);
                }
            };
            if (PF) frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int cur = PF ? (ks & 1) : 0;
                if (PF) { if (ks + 1 < 4) frags(ks + 1, cur ^ 1); }
                else frags(ks, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                    if (PF == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (have_next) piece(ks * 4 + i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            stage_no++;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc[i][j][e];
    if (s == 12345.678f) out[tid] = s;          // (keeps the accumulators alive)
}

// the same tile with FOUR 32-KB stages of BK = 32 (three in flight, 96 KB) instead of two 64-KB stages (one in flight, 64 KB)
__device__ __forceinline__ int swz4(int row) { return (row >> 2) & 3; }
__global__ __launch_bounds__(256) void proto_deep(const bf16 *__restrict__ X, const bf16 *__restrict__ W, float *__restrict__ out, int M, int N, int K,
                                                  int jobs_per_wg, int map)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T2 = 256 * 64, ST2 = 2 * T2;           // operand tile 256 rows x 32 bf16 = 16 KB; stage 32 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = M / 256, tiles_n = N / 256, nk = K / 32;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    unsigned vo[4];                                     // this wave's 4 + 4 one-KiB pieces per stage: piece q = rows 16q .. 16q+15
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 16 * (4 * wave + i) + (lane >> 2), ps = lane & 3;
        vo[i] = (unsigned)((row * K + ((ps ^ swz4(row)) << 3)) * 2);
    }
    const int total = jobs_per_wg * nk;                 // stages of this workgroup, all jobs in sequence
    auto issue = [&](int s) {                           // stage s (global index) -> buffer s & 3
        const int jj = s / nk, kt = s - jj * nk;
        const int job = blockIdx.x + jj * gridDim.x;
        int m0 = (job % tiles_m) * 256, n0 = ((job / tiles_m) % tiles_n) * 256;
        if (map) { const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3; m0 = ((xcd * 4 + (i & 3) + jj * 32) % tiles_m) * 256; n0 = ((i >> 2) % tiles_n) * 256; }
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)m0 * K), 0, 256 * K * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void *)(W + (size_t)n0 * K), 0, 256 * K * 2, 0x00020000);
        unsigned char *base = smem + (s & 3) * ST2 + (4 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void *)(base + i * 1024), 16, vo[i], kt * 64, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void *)(base + T2 + i * 1024), 16, vo[i], kt * 64, 0, 0);
        }
    };
    issue(0); issue(1); issue(2);
    for (int s = 0; s < total; s++) {
        if (s + 2 < total) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 3 < total) issue(s + 3);
        const unsigned char *As = smem + (s & 3) * ST2 + (wr * 128) * 64;
        const unsigned char *Bs = smem + (s & 3) * ST2 + T2 + (wc * 128) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 32 * i + r;
                const int off = row * 64 + (((2 * ks + hh) ^ swz4(row)) << 4);
                a[i] = *reinterpret_cast<const bf16x8 *>(As + off);
                b[i] = *reinterpret_cast<const bf16x8 *>(Bs + off);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) sum += acc[i][j][e];
    if (sum == 12345.678f) out[tid] = sum;
}

int main()
{
    const int M = 87904 / 256 * 256, N = 2304, K = 768;
    bf16 *X, *W;
    float *out;
    hipMalloc(&X, (size_t)M * K * 2);
    hipMalloc(&W, (size_t)N * K * 2);
    hipMalloc(&out, 4096);
    hipMemset(X, 0, (size_t)M * K * 2);
    hipMemset(W, 0, (size_t)N * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int jobs = 12, nk = K / BK;
    for (int pf = 0; pf < 3; pf++) {
        auto kern = pf == 2 ? proto<2> : (pf ? proto<1> : proto<0>);
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        for (int rep = 0; rep < 4; rep++) {
            const int map = rep >= 2;
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256), 2 * STAGE, 0, X, W, out, M, N, K, jobs, map);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double per_tile_us = ms * 1e3 / (jobs * nk);
            const double tf = 2.0 * 256 * 256 * BK * 256.0 * jobs * nk / (ms * 1e-3) / 1e12;
            printf("prefetch=%d xcd-map=%d rep %d: %.3f ms  %.3f us per K-tile  %.0f TFLOP/s in the loop (%s)\n", pf, map, rep, ms, per_tile_us, tf, hipGetErrorString(hipGetLastError()));
        }
    }
    hipFuncSetAttribute((const void *)proto_deep, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 256 * 64);
    for (int rep = 0; rep < 4; rep++) {
        const int map = rep >= 2;
        hipEventRecord(e0);
        hipLaunchKernelGGL(proto_deep, dim3(256), dim3(256), 4 * 2 * 256 * 64, 0, X, W, out, M, N, K, jobs, map);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("deep ring (4 x 32 KB) xcd-map=%d rep %d: %.3f ms  %.3f us per 64-wide K-tile  %.0f TFLOP/s in the loop (%s)\n", map, rep, ms, ms * 1e3 / (jobs * nk),
               2.0 * 256 * 256 * BK * 256.0 * jobs * nk / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
