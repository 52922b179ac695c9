// radix_sort.hpp -- stable LSD radix sort of (32-bit key, 32-bit value) pairs on the device, 8 bits per pass (round 4: replaces
// rocprim::radix_sort_pairs, the last library primitive on the training path).
//
// Used once per lattice build (permuto_kernels.hip): the 6 N (vertex, pixel) pairs of a batch of images are ordered by (image, vertex) so that
// every lattice vertex finds its pairs as one run IN PIXEL ORDER -- the order of the reference's serial splat loop
// (utils/bilateralfilter/permutohedral.cpp:507-530), which is what makes the filtered planes bit-identical to the reference's.  So the sort
// must be STABLE; it is an all-integer counting sort per digit, hence also the same bits every run.
//
// One pass over digit d = (key >> shift) & 255, tiles of 2048 consecutive pairs per 256-thread workgroup:
//   rs_hist_kernel     counts[d][tile] = number of pairs of the tile with digit d                      (LDS histogram)
//   rs_rowscan_kernel  per digit: exclusive scan of its row over the tiles + the row total             (256 workgroups, coalesced)
//   rs_scatter_kernel  stable rank of every pair inside its tile (wave w owns the tile's pairs [512 w, 512 w + 512) in eight slots of 64
//                      consecutive pairs; the lanes of a slot with equal digits find each other with eight ballots, a per-wave LDS counter per
//                      digit carries the rank from slot to slot), local reorder through LDS, then a coalesced write-out:
//                      out[ digit base + row scan(tile) + rank in tile ]
// Tile order x in-tile order = input order for equal digits: stable.  Scratch: (256 * ntiles + 256) * 4 bytes.
#pragma once
#include <hip/hip_runtime.h>

namespace cosa {
namespace {

constexpr int RS_TILE = 2048, RS_THREADS = 256;

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned *__restrict__ keys, size_t n, int shift, unsigned *__restrict__ counts, int ntiles)
{
    __shared__ unsigned hist[256];
    const int tid = threadIdx.x, tile = blockIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)tile * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_TILE / RS_THREADS; j++) {
        const size_t i = base + (size_t)j * RS_THREADS + tid;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)tid * ntiles + tile] = hist[tid];
}

// row d of counts: exclusive scan over the tiles in place, total -> rowsum[d]
__global__ __launch_bounds__(RS_THREADS) void rs_rowscan_kernel(unsigned *__restrict__ counts, unsigned *__restrict__ rowsum, int ntiles)
{
    __shared__ unsigned part[RS_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned *row = counts + (size_t)blockIdx.x * ntiles;
    unsigned carry = 0;
    for (int c0 = 0; c0 < ntiles; c0 += RS_THREADS) {
        const int i = c0 + tid;
        const unsigned v = i < ntiles ? row[i] : 0u;
        unsigned s = v;                                        // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) part[wave] = s;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; w++) woff += part[w];
        const unsigned total = part[0] + part[1] + part[2] + part[3];
        if (i < ntiles) row[i] = carry + woff + s - v;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) rowsum[blockIdx.x] = carry;
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const unsigned *__restrict__ keys_in, const unsigned *__restrict__ vals_in,
                                                                unsigned *__restrict__ keys_out, unsigned *__restrict__ vals_out, size_t n, int shift,
                                                                const unsigned *__restrict__ counts, const unsigned *__restrict__ rowsum, int ntiles)
{
    __shared__ unsigned cntw[4][256];         // per wave and digit: pairs seen so far; later: pairs of the earlier waves
    __shared__ unsigned lstart[256];          // first local position of a digit's run in the reordered tile
    __shared__ unsigned gdelta[256];          // global index = local position + gdelta[digit]
    __shared__ unsigned scan_tmp[256];
    __shared__ unsigned skey[RS_TILE], sval[RS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    volatile unsigned *cw = &cntw[wave][0];
#pragma unroll
    for (int w = 0; w < 4; w++) cntw[w][tid] = 0;
    __syncthreads();
    const size_t base = (size_t)tile * RS_TILE + (size_t)wave * 512;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned key[8], val[8], rank[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const size_t i = base + (size_t)s * 64 + lane;
        const bool valid = i < n;
        key[s] = valid ? keys_in[i] : 0xffffffffu;
        val[s] = valid ? vals_in[i] : 0u;
        const unsigned d = (key[s] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned before = valid ? cw[d] : 0u;
        const unsigned lower = (unsigned)__popcll(peers & lt);
        rank[s] = before + lower;
        if (valid && lower == 0) cw[d] = before + (unsigned)__popcll(peers);          // the run's first lane books the whole run
    }
    __syncthreads();
    {       // thread = digit: offsets of the waves inside the digit's run, the run's start in the tile, and its place in the output
        const unsigned c0 = cntw[0][tid], c1 = cntw[1][tid], c2 = cntw[2][tid], c3 = cntw[3][tid];
        cntw[0][tid] = 0; cntw[1][tid] = c0; cntw[2][tid] = c0 + c1; cntw[3][tid] = c0 + c1 + c2;
        const unsigned tot = c0 + c1 + c2 + c3, rs = rowsum[tid];
        // two exclusive scans over the 256 digits: the tile's run starts and the digits' bases in the output
        unsigned a = tot, g = rs;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned ta = __shfl_up(a, o, 64), tg = __shfl_up(g, o, 64);
            if (lane >= o) { a += ta; g += tg; }
        }
        if (lane == 63) { scan_tmp[wave] = a; scan_tmp[4 + wave] = g; }
        __syncthreads();
        unsigned wa = 0, wg = 0;
        for (int w = 0; w < wave; w++) { wa += scan_tmp[w]; wg += scan_tmp[4 + w]; }
        const unsigned ls = wa + a - tot, gb = wg + g - rs;
        lstart[tid] = ls;
        gdelta[tid] = gb + counts[(size_t)tid * ntiles + tile] - ls;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const size_t i = base + (size_t)s * 64 + lane;
        if (i < n) {
            const unsigned d = (key[s] >> shift) & 255u;
            const unsigned lp = lstart[d] + cntw[wave][d] + rank[s];
            skey[lp] = key[s];
            sval[lp] = val[s];
        }
    }
    __syncthreads();
    const size_t tbase = (size_t)tile * RS_TILE;
    const int count = (int)(n - tbase < (size_t)RS_TILE ? n - tbase : (size_t)RS_TILE);
    for (int j = tid; j < count; j += RS_THREADS) {
        const unsigned k = skey[j];
        const unsigned out = (unsigned)j + gdelta[(k >> shift) & 255u];
        keys_out[out] = k;
        vals_out[out] = sval[j];
    }
}

inline size_t rs_scratch_bytes(size_t n)
{
    const size_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    return (256 * ntiles + 256) * sizeof(unsigned);
}

// sorts n pairs by the low `bits` bits of the key, ping-ponging between the two buffer pairs; returns 0 / 1: which pair holds the result
// (-1: a launch failed).  n < 2^32.
inline int rs_sort_pairs(unsigned *k0, unsigned *v0, unsigned *k1, unsigned *v1, size_t n, int bits, void *scratch, hipStream_t st)
{
    const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
    unsigned *counts = static_cast<unsigned *>(scratch), *rowsum = counts + (size_t)256 * ntiles;
    unsigned *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    int where = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(ntiles), dim3(RS_THREADS), 0, st, ki, n, shift, counts, ntiles);
        hipLaunchKernelGGL(rs_rowscan_kernel, dim3(256), dim3(RS_THREADS), 0, st, counts, rowsum, ntiles);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(ntiles), dim3(RS_THREADS), 0, st, ki, vi, ko, vo, n, shift, counts, rowsum, ntiles);
        if (hipGetLastError() != hipSuccess) return -1;
        unsigned *t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
        where ^= 1;
    }
    return where;
}

}  // namespace
}  // namespace cosa
