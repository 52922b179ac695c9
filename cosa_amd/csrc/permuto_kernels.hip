// permuto_kernels.hip -- permutohedral-lattice bilateral filter + DenseEnergyLoss core on gfx950.
//
// Reference: utils/bilateralfilter/bilateralfilter.cpp:4-55 (5-D features, batch driver),
//            utils/bilateralfilter/permutohedral.cpp:115-297 (lattice init, SSE path: blocks of 4,
//            round-to-nearest-even), :507-571 (splat / 6 blur passes / slice, value_size = 1),
//            utils/seg_helper.py:864-903 (DenseEnergyLossFunction).
//
// MI355X design (not the reference's serial hash + K scalar passes):
//   * one lattice per image, built on device: every pixel computes its 6 simplex vertices
//     (same op order as the reference => identical keys and barycentric weights) and inserts
//     64-bit packed keys into an open-addressing table with atomicCAS; the winner of a slot draws
//     the dense lattice id.  No host round trip: M stays on the device, launches are sized for
//     the worst case (6*N) and read M to exit early.
//   * all K channels ride through the lattice together: values[M+1][KP] rows (channel fastest),
//     so splat / blur / slice touch whole rows instead of re-streaming the lattice K times.
//   * planar [K][H*W] <-> row [pixel][KP] transposes go through LDS tiles.
//   * every kernel gathers rows of ONE image's tables in an order no cache line survives for long, and workgroup ids are dealt round-robin
//     to the 8 XCDs (private 4-MB L2 each): with an image's workgroups on all XCDs every table is fetched into eight L2s.  image_wg()
//     maps a 1-D grid to (image, workgroup of the image) so that an image lives on ONE XCD (batch >= 8; smaller batches split an image
//     over 8 / N XCDs) -- the same work, the same bits, an eighth of the fabric traffic (round 5).
//   * a hash-table entry is 16 bytes {packed key, dense id}: a probe that finds its key has the id in the same cache line.
//   * splat without float atomics: the (pixel, vertex) pairs of an image are sorted by vertex once per lattice (stable LSD radix sort,
//     radix_sort.hpp; the pairs are generated in pixel order, so a vertex's list is in ascending pixel order), and a
//     half-wave per vertex adds its list up in that order -- the order of the reference's serial splat loop (permutohedral.cpp:507-530),
//     so the value rows are bit-identical to the CPU's and the same from run to run.
// Compiled with -ffp-contract=off (lattice coordinates must match the CPU oracle bit for bit).
#include "kernels.hpp"
#include <cmath>
#include <cstring>
#include <vector>
#include <type_traits>
#include "radix_sort.hpp"

namespace cosa {
namespace {

// The lattice is dimension-generic.  COSA_PD = 5 (default build): positions / sigma_xy + colours / sigma_rgb -- the bilateral filter of the
// hot path.  The same file compiled with -DCOSA_PD=2 (object permuto_kernels_d2.o, entry points cosa_lattice_filter_d2*) is the
// position-only Gaussian kernel of the dense-CRF post-processing (utils/seg_helper.py:961-996: pydensecrf's addPairwiseGaussian).
#ifndef COSA_PD
#define COSA_PD 5
#endif
constexpr int PD = COSA_PD;
constexpr int PD1 = COSA_PD + 1;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int TP = 64;  // pixels per LDS transpose tile
constexpr int kAllXcds = 8;        // image_wg(): `parts` of a kernel whose images are shared by all XCDs
constexpr int kWgPerImage = 1024;  // workgroups per image of the grid-stride kernels (they read M on the device; an image's XCD holds 256 at a time)

struct LatticeParams {
    float scale[PD];   // diag of E: 1/sqrt((i+1)(i+2)) * sqrt(2/3)*(d+1)
    float inv_sxy, inv_srgb_unused;
    float sigmaxy, sigmargb;
    int H, W, N, Npad;
    unsigned cap_mask;  // table capacity - 1 (power of two)
    int Mmax;           // rows reserved per image (excluding sink row 0)
    int KP;             // padded channel count
    int nimg;           // images in the batch
    int parts;          // XCDs that share one image (1 when the batch has >= 8 images): image_wg()
};

struct TableEntry {     // one slot of an image's open-addressing table (kEmpty key: free)
    unsigned long long key;
    int id;             // dense lattice id of the key (valid after lattice_build_kernel)
    int pad;
};

struct ImageBuffers {   // per-image strides (in elements) into the workspace arrays
    TableEntry *table;          // [N][cap]
    unsigned long long *pkey;   // [N][Mmax]
    int *offset;                // [N][Npad*6]
    float *bary;                // [N][Npad*6]
    int2 *nb;                   // [N][6][Mmax]
    float *val0, *val1;         // [N][(Mmax+1)*KP]
    int *M;                     // [N]
    int *err;                   // [1]
    double *loss_acc;           // [1]
    // sorted splat lists: pair e = 6 * pixel + r of image n has key (n << id_bits) | vertex id (padding pixels: id = Mmax)
    unsigned *ckey0, *ckey1;    // [N*Npad*6] keys before / after the sort
    unsigned *cent0, *cent1;    // [N*Npad*6] pair numbers e before / after the sort
    int *seg_lo, *seg_hi;       // [N][Mmax] range of a vertex's pairs in cent1 (lo == hi == 0: none)
    float *rows;                // [N][Npix][KP] the input of a filter pass as pixel rows (x roi)
    double *loss_part;          // [N * ceil(Npix / TP)] per-workgroup partial sums of the energy
    void *sort_tmp;             // the radix sort's digit counts (radix_sort.hpp)
    size_t sort_tmp_bytes;
    int id_bits;
};

__device__ __forceinline__ unsigned long long hmix(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

// pack 5 lattice coordinates (all congruent mod 6) into 63 bits: r | 5 x 12-bit quotients
__device__ __forceinline__ bool pack_key(const int *key, unsigned long long &pk)
{
    int r = key[0] % PD1;
    if (r < 0) r += PD1;
    pk = (unsigned long long)r;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < PD; i++) {
        const int q = (key[i] - r) / PD1;
        ok = ok && (q >= -2048) && (q <= 2047);
        pk |= (unsigned long long)((unsigned)(q + 2048) & 0xFFFu) << (3 + 12 * i);
    }
    return ok;
}

// dense id of a packed key (-1: not a lattice point)
__device__ __forceinline__ int find_id(const TableEntry *table, unsigned mask, unsigned long long pk)
{
    unsigned s = (unsigned)hmix(pk) & mask;
    for (;;) {
        const TableEntry t = table[s];          // (one 16-byte load)
        if (t.key == pk) return t.id;
        if (t.key == kEmpty) return -1;
        s = (s + 1) & mask;
    }
}

// 1-D grid -> (image n, workgroup l of the image's per_image workgroups), all workgroups of a unit (an image, or one of `parts` interleaved
// slices of its workgroups) on one XCD: workgroup id i runs on XCD i % 8.  Unit u = n * parts + slice lives on XCD u % 8; the units of an
// XCD run one after the other.  Grid size: image_grid().
__device__ __forceinline__ bool image_wg(int N, int parts, int per_image, int &n, int &l)
{
    const int lp = (per_image + parts - 1) / parts;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int u = xcd + 8 * (j / lp);
    if (u >= N * parts) return false;
    n = u / parts;
    l = (j % lp) * parts + (u - n * parts);
    return l < per_image;
}

// ---- 1. per-pixel simplex + hash insertion ----------------------------------------------------
__global__ __launch_bounds__(256) void lattice_build_kernel(const float *__restrict__ images, LatticeParams P, ImageBuffers B)
{
    int n, wg;
    if (!image_wg(P.nimg, P.parts, (P.Npad + 255) / 256, n, wg)) return;
    const int p = wg * 256 + threadIdx.x;
    if (p >= P.Npad) return;
    const size_t hw = (size_t)P.H * P.W;
    const float *img = images + (size_t)n * 3 * hw;
    float f[PD];
#pragma unroll
    for (int i = 0; i < PD; i++) f[i] = 0.f;
    if (p < P.N) {
        const int yj = p / P.W, xi = p - yj * P.W;
        f[0] = (float)xi / P.sigmaxy;
        f[1] = (float)yj / P.sigmaxy;
#if COSA_PD == 5
        f[2] = img[p] / P.sigmargb;
        f[3] = img[hw + p] / P.sigmargb;
        f[4] = img[2 * hw + p] / P.sigmargb;
#else
        (void)img;
#endif
    }
    float el[PD1], rem0[PD1], rank[PD1], bc[PD1 + 1];
    float sm = 0.0f;
#pragma unroll
    for (int j = PD; j > 0; j--) {
        const float cf = f[j - 1] * P.scale[j - 1];
        el[j] = sm - (float)j * cf;
        sm = sm + cf;
    }
    el[0] = sm;
    const float inv6 = 1.0f / (float)PD1;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i <= PD; i++) {
        float v = inv6 * el[i];
        v = __builtin_rintf(v);
        rem0[i] = v * (float)PD1;
        sum = sum + v;
        rank[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < PD; i++) {
        const float di = el[i] - rem0[i];
#pragma unroll
        for (int j = i + 1; j <= PD; j++) {
            const float dj = el[j] - rem0[j];
            if (di < dj) rank[i] = rank[i] + 1.0f; else rank[j] = rank[j] + 1.0f;
        }
    }
#pragma unroll
    for (int i = 0; i <= PD; i++) {
        rank[i] = rank[i] + sum;
        if (rank[i] < 0.0f) { rank[i] = rank[i] + (float)PD1; rem0[i] = rem0[i] + (float)PD1; }
        else if (rank[i] >= (float)PD1) { rank[i] = rank[i] - (float)PD1; rem0[i] = rem0[i] - (float)PD1; }
    }
#pragma unroll
    for (int i = 0; i <= PD + 1; i++) bc[i] = 0.0f;
    // scatter into bc[d-rank], bc[d-rank+1] with static indexing (keeps bc in registers)
#pragma unroll
    for (int i = 0; i <= PD; i++) {
        const float v = (el[i] - rem0[i]) * inv6;
        const int q = PD - (int)rank[i];
#pragma unroll
        for (int s = 0; s <= PD + 1; s++) {
            if (s == q) bc[s] = bc[s] + v;
            if (s == q + 1) bc[s] = bc[s] - v;
        }
    }
    bc[0] = bc[0] + (1.0f + bc[PD + 1]);

    TableEntry *table = B.table + (size_t)n * ((size_t)P.cap_mask + 1);
    unsigned long long *pkey = B.pkey + (size_t)n * P.Mmax;
    int *Mp = B.M + n;
    bool bad = false;
    // Three stages, each over the six vertices r, so that a wave's six table probes are in flight TOGETHER (round 5: one after the other they
    // were six load latencies -- plus a CAS latency for every new point -- per wave):
    //   1. keys, and wave-level de-duplication: on smooth images the 64 consecutive pixels of a wave share a handful of lattice points per r,
    //      so one lane per distinct key (the first that holds it) probes / inserts and the others take its slot by a cross-lane read: an
    //      order of magnitude fewer random accesses into the table.  The grouping loop is ballots and scalar compares only.
    //   2. the leaders' first probes (plain loads), all six issued back to back;
    //   3. test before test-and-set: a slot only ever goes kEmpty -> key, so a plain (possibly stale) load can at worst still show kEmpty, in
    //      which case the CAS decides: most attempts end with the load of stage 2 instead of a memory-side 64-bit atomic on a contended
    //      address.  Tried and dropped: de-duplicating a workgroup's 1536 keys in an LDS hash first (1.20 ms) and pre-aggregating the splat
    //      contributions of a workgroup in LDS rows (0.75 vs 0.71 ms) -- the occupancy lost to the LDS tables cost more than they saved.
    // new_slot / new_key / won: slots this lane claimed (a new lattice point): the dense ids are handed out after the loops.
    const int lane = threadIdx.x & 63;
    unsigned new_slot[PD1];
    unsigned long long new_key[PD1], seen[PD1];
    int leader[PD1];
    bool won[PD1];
#pragma unroll
    for (int r = 0; r <= PD; r++) {
        int key[PD];
#pragma unroll
        for (int i = 0; i < PD; i++) {
            // canonical[r][rank] = r if rank <= d-r else r-(d+1)
            const int rk = (int)rank[i];
            const float can = (float)(rk <= PD - r ? r : r - PD1);
            key[i] = (int)(rem0[i] + can);
        }
        unsigned long long pk;
        if (!pack_key(key, pk)) bad = true;
        const unsigned pk_lo = (unsigned)pk, pk_hi = (unsigned)(pk >> 32);
        int ld = lane;
        for (unsigned long long rem = __ballot(1); rem != 0;) {
            const int l = __ffsll((long long)rem) - 1;                                 // (wave-uniform)
            const unsigned llo = __builtin_amdgcn_readlane(pk_lo, l), lhi = __builtin_amdgcn_readlane(pk_hi, l);
            const bool mine = pk_lo == llo && pk_hi == lhi;
            if (mine) ld = l;                                                          // (a later round cannot match again: the key is gone from rem)
            rem &= ~__ballot(mine);
        }
        leader[r] = ld;
        new_key[r] = pk;
        new_slot[r] = (unsigned)hmix(pk) & P.cap_mask;
    }
#pragma unroll
    for (int r = 0; r <= PD; r++) {
        seen[r] = kEmpty;
        if (leader[r] == lane) seen[r] = __builtin_nontemporal_load(&table[new_slot[r]].key);
    }
#pragma unroll
    for (int r = 0; r <= PD; r++) {
        won[r] = false;
        if (leader[r] == lane) {
            const unsigned long long pk = new_key[r];
            unsigned s = new_slot[r];
            unsigned long long sn = seen[r];
            for (;;) {
                if (sn == pk) break;
                if (sn == kEmpty) {
                    const unsigned long long prev = atomicCAS(&table[s].key, kEmpty, pk);
                    if (prev == kEmpty) { won[r] = true; break; }
                    if (prev == pk) break;
                }
                s = (s + 1) & P.cap_mask;
                sn = __builtin_nontemporal_load(&table[s].key);
            }
            new_slot[r] = s;
        }
    }
    {       // slot for now (remapped to the dense id next) and barycentric weights: 24 + 24 bytes per pixel, 8-byte stores
        int so[PD1];
#pragma unroll
        for (int r = 0; r <= PD; r++) so[r] = __shfl((int)new_slot[r], leader[r], 64);
        int *op = B.offset + ((size_t)n * P.Npad + p) * PD1;
        float *bp = B.bary + ((size_t)n * P.Npad + p) * PD1;
#if COSA_PD == 5
#pragma unroll
        for (int r = 0; r < PD1; r += 2) {
            *reinterpret_cast<int2 *>(op + r) = make_int2(so[r], so[r + 1]);
            *reinterpret_cast<float2 *>(bp + r) = make_float2(bc[r], bc[r + 1]);
        }
#else
#pragma unroll
        for (int r = 0; r <= PD; r++) { op[r] = so[r]; bp[r] = bc[r]; }
#endif
    }
    // Dense ids: ONE add to the image's counter per wave for all the points its lanes created (ranked by ballot), not one per point:
    // the per-image counter is a single address, and same-address atomics serialise at ~20 ns each -- with one add per new point they
    // were 430 of this kernel's 630 us (b = 16, 224^2, ~25 k points per image).
    {
        unsigned long long wm[PD1];
        int total = 0;
#pragma unroll
        for (int r = 0; r <= PD; r++) { wm[r] = __ballot(won[r]); total += __popcll(wm[r]); }
        if (total) {                                                                   // (wave-uniform)
            const int first = __ffsll((long long)__ballot(1)) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(Mp, total);
            base = __shfl(base, first, 64);
#pragma unroll
            for (int r = 0; r <= PD; r++) {
                if (won[r]) {
                    const int id = base + __popcll(wm[r] & ((1ull << lane) - 1ull));
                    table[new_slot[r]].id = id;
                    pkey[id] = new_key[r];
                }
                base += __popcll(wm[r]);
            }
        }
    }
    if (bad) atomicExch(B.err, 1);
}

// ---- 2a. slot -> dense id ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lattice_remap_kernel(LatticeParams P, ImageBuffers B)
{
    int n, wg;
    if (!image_wg(P.nimg, P.parts, (P.Npad * PD1 + 255) / 256, n, wg)) return;
    const int i = wg * 256 + threadIdx.x;
    if (i >= P.Npad * PD1) return;
    int *off = B.offset + (size_t)n * P.Npad * PD1;
    const TableEntry *table = B.table + (size_t)n * ((size_t)P.cap_mask + 1);
    const int id = table[off[i]].id;
    off[i] = id;
    const size_t g = (size_t)n * P.Npad * PD1 + i;
    B.ckey0[g] = ((unsigned)n << B.id_bits) | (unsigned)(i / PD1 < P.N ? id : P.Mmax);
    B.cent0[g] = (unsigned)i;
}

// ---- 2c. ranges of the sorted (vertex, pair) list ---------------------------------------------------
__global__ __launch_bounds__(256) void lattice_bounds_kernel(size_t total, LatticeParams P, ImageBuffers B)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned k = B.ckey1[i], mask = (1u << B.id_bits) - 1u;
    const unsigned id = k & mask, n = k >> B.id_bits;
    if (id >= (unsigned)P.Mmax) return;
    if (i == 0 || B.ckey1[i - 1] != k) B.seg_lo[(size_t)n * P.Mmax + id] = (int)i;
    if (i + 1 == total || B.ckey1[i + 1] != k) B.seg_hi[(size_t)n * P.Mmax + id] = (int)(i + 1);
}

// ---- 2b. blur neighbours + zero the value rows -----------------------------------------------------
__device__ __forceinline__ void unpack_key(unsigned long long pk, int *key)
{
    const int r = (int)(pk & 7ull);
#pragma unroll
    for (int i = 0; i < PD; i++) key[i] = ((int)((pk >> (3 + 12 * i)) & 0xFFFull) - 2048) * PD1 + r;
}

__global__ __launch_bounds__(256) void lattice_neighbors_kernel(int per_image, LatticeParams P, ImageBuffers B)
{
    int n, wg;
    if (!image_wg(P.nimg, P.parts, per_image, n, wg)) return;
    const int M = B.M[n];
    const TableEntry *table = B.table + (size_t)n * ((size_t)P.cap_mask + 1);
    const unsigned long long *pkey = B.pkey + (size_t)n * P.Mmax;
    int2 *nb = B.nb + (size_t)n * PD1 * P.Mmax;
    for (int e = wg * 256 + threadIdx.x; e < M * PD1; e += per_image * 256) {
        const int j = e / M, i = e - j * M;
        int key[PD], k1[PD], k2[PD];
        unpack_key(pkey[i], key);
#pragma unroll
        for (int k = 0; k < PD; k++) { k1[k] = key[k] - 1; k2[k] = key[k] + 1; }
#pragma unroll
        for (int k = 0; k < PD; k++)
            if (k == j) { k1[k] = key[k] + PD; k2[k] = key[k] - PD; }
        unsigned long long p1, p2;
        int r1 = -1, r2 = -1;
        if (pack_key(k1, p1)) r1 = find_id(table, P.cap_mask, p1);
        if (pack_key(k2, p2)) r2 = find_id(table, P.cap_mask, p2);
        nb[(size_t)j * P.Mmax + i] = make_int2(r1, r2);
    }
}

// ---- 3. splat: values[id+1][k] += bary * in[k][p]  (optionally in = seg*roi), without atomics ---------
// rows[n][p][k] = in[n][k][p] (* roi[n][p]): the filter input as pixel rows (LDS tile transpose)
__global__ __launch_bounds__(256) void lattice_rows_kernel(const float *__restrict__ ins, const float *__restrict__ roi,
                                                          int K, LatticeParams P, ImageBuffers B)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [TP][KP+1]
    int n, wg;
    if (!image_wg(P.nimg, kAllXcds, (P.N + TP - 1) / TP, n, wg)) return;          // (a streaming kernel: nothing to keep in one L2)
    const int p0 = wg * TP;
    const int KP = P.KP, ld = KP + 1;
    const size_t hw = (size_t)P.N;
    const float *in = ins + (size_t)n * K * hw;
    for (int e = threadIdx.x; e < TP * KP; e += 256) {
        const int k = e / TP, pl = e - k * TP;
        float v = 0.0f;
        if (k < K && p0 + pl < P.N) {
            v = in[(size_t)k * hw + p0 + pl];
            if (roi) v = v * roi[(size_t)n * hw + p0 + pl];
        }
        tile[pl * ld + k] = v;
    }
    __syncthreads();
    float *rows = B.rows + ((size_t)n * hw + p0) * KP;
    for (int e = threadIdx.x; e < TP * KP; e += 256) {
        const int pl = e / KP, k = e - pl * KP;
        if (p0 + pl < P.N) rows[e] = tile[pl * ld + k];
    }
}

// values[id+1][k] = sum over the vertex's pairs, in ascending pixel order, of bary * rows[pixel][k]: 32 lanes per vertex (lane = channel).
// The half-wave first fetches up to 32 of the vertex's pairs TOGETHER (lane u: pair number and barycentric weight of pair u -- two dependent
// loads for 32 pairs instead of two per pair), then walks them in order with the pair broadcast by a lane shuffle, so the row loads have no
// load in front of them and several are in flight (round 5: lists average 12 pairs, the kernel was a chain of load latencies).
// Also zeroes the sink rows (row 0 of both value buffers).
template <int NT>       // NT 32-channel groups per lane: KP <= 32 NT (VOC 21 planes: 1; COCO 81: 3) -- a vertex's pair list is walked once
__global__ __launch_bounds__(256) void lattice_splat_sorted_kernel(int per_image, LatticeParams P, ImageBuffers B)
{
    int n, wg;
    if (!image_wg(P.nimg, P.parts, per_image, n, wg)) return;
    const int M = B.M[n];
    const int KP = P.KP;
    const int k = threadIdx.x & 31;
    const size_t vstride = ((size_t)P.Mmax + 1) * KP;
    float *val = B.val0 + (size_t)n * vstride;
    if (wg == 0) {
        for (int c = threadIdx.x; c < KP; c += 256) {
            val[c] = 0.0f;
            B.val1[(size_t)n * vstride + c] = 0.0f;
        }
    }
    const int *lo = B.seg_lo + (size_t)n * P.Mmax, *hi = B.seg_hi + (size_t)n * P.Mmax;
    const float *bary = B.bary + (size_t)n * P.Npad * PD1;
    const float *rows = B.rows + (size_t)n * P.N * KP;
    bool on[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) on[t] = k + 32 * t < KP;
    for (int id = wg * 8 + (threadIdx.x >> 5); id < M; id += per_image * 8) {
        const int beg = lo[id], end = hi[id];
        for (int kc0 = 0; kc0 < KP; kc0 += 32 * NT) {     // (one trip unless KP > 32 NT)
            float acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = 0.0f;
            // a group of U pairs: U row loads in flight, then the U multiply-adds in pair order.  Long lists (a lattice point of a flat image
            // region collects thousands of pixels) are one latency per GROUP, and the kernel ends with its longest list: groups of 16
            auto group = [&](const unsigned e_l, const float w_l, const int i0, auto utag) {
                constexpr int U = decltype(utag)::value;
                float w[U], v[U][NT];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned e = (unsigned)__shfl((int)e_l, i0 + u, 32);
                    w[u] = __shfl(w_l, i0 + u, 32);
                    const float *rp = rows + (size_t)(e / PD1) * KP + kc0 + k;
#pragma unroll
                    for (int t = 0; t < NT; t++) v[u][t] = (on[t] && kc0 + k + 32 * t < KP) ? rp[32 * t] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t] = acc[t] + w[u] * v[u][t];
            };
            constexpr int UB = NT == 1 ? 16 : 8;
            unsigned e_n = 0u;
            float w_n = 0.0f;
            if (beg + k < end) {
                e_n = B.cent1[beg + k];
                w_n = bary[e_n];
            }
            for (int base = beg; base < end; base += 32) {
                const int cnt = end - base < 32 ? end - base : 32;          // (the same for the 32 lanes of the vertex)
                const unsigned e_l = e_n;
                const float w_l = w_n;
                if (base + 32 + k < end) {                                  // the next 32 pairs: in flight under this chunk's rows
                    e_n = B.cent1[base + 32 + k];
                    w_n = bary[e_n];
                }
                int i = 0;
                for (; i + UB <= cnt; i += UB) group(e_l, w_l, i, std::integral_constant<int, UB>{});
                for (; i + 4 <= cnt; i += 4) group(e_l, w_l, i, std::integral_constant<int, 4>{});
                for (; i < cnt; i++) group(e_l, w_l, i, std::integral_constant<int, 1>{});
            }
#pragma unroll
            for (int t = 0; t < NT; t++)
                if (kc0 + k + 32 * t < KP) val[(size_t)(id + 1) * KP + kc0 + k + 32 * t] = acc[t];
        }
    }
}

// ---- 4. blur along axis j: new = old + 0.5*(old[n1] + old[n2]) ----------------------------------------
__global__ __launch_bounds__(256) void lattice_blur_kernel(int axis, int parity, int per_image, LatticeParams P, ImageBuffers B)
{
    // all XCDs share every image: the value rows of an image (2 x 2.4 MB at M = 25 k, K = 21) fit any L2, and the images' lattice sizes differ --
    // with an image per XCD the pass waits for the XCD that drew the largest ones (measured: 17.6 us shared, 18.9-20.4 us per XCD)
    int n, wg;
    if (!image_wg(P.nimg, kAllXcds, per_image, n, wg)) return;
    const int M = B.M[n];
    const int KP = P.KP, q4 = KP / 4;
    const size_t vstride = ((size_t)P.Mmax + 1) * KP;
    const float *oldv = (parity ? B.val1 : B.val0) + (size_t)n * vstride;
    float *newv = (parity ? B.val0 : B.val1) + (size_t)n * vstride;
    const int2 *nb = B.nb + ((size_t)n * PD1 + axis) * P.Mmax;
    const int tot = M * q4;
    for (int e = wg * 256 + threadIdx.x; e < tot; e += per_image * 256) {
        const int i = e / q4, q = e - i * q4;
        const int2 nn = nb[i];
        const float4 o = *reinterpret_cast<const float4 *>(oldv + (size_t)(i + 1) * KP + 4 * q);
        const float4 a = *reinterpret_cast<const float4 *>(oldv + (size_t)(nn.x + 1) * KP + 4 * q);
        const float4 b = *reinterpret_cast<const float4 *>(oldv + (size_t)(nn.y + 1) * KP + 4 * q);
        float4 r;
        r.x = o.x + 0.5f * (a.x + b.x);
        r.y = o.y + 0.5f * (a.y + b.y);
        r.z = o.z + 0.5f * (a.z + b.z);
        r.w = o.w + 0.5f * (a.w + b.w);
        *reinterpret_cast<float4 *>(newv + (size_t)(i + 1) * KP + 4 * q) = r;
    }
}

// ---- 5. slice (+ optional DenseEnergy gate / loss) ---------------------------------------------------
// out[k][p] = sum_r (bary*alpha) * val[id_r+1][k]
// energy mode: gate = clamp(roi - max_k seg, 0), gate[unlabel] = 1; out *= gate; loss += seg*roi*out
__global__ __launch_bounds__(256) void lattice_slice_kernel(float *__restrict__ outs, int K, int final_parity,
                                                           const float *__restrict__ seg, const float *__restrict__ roi,
                                                           const unsigned char *__restrict__ unlabel,
                                                           LatticeParams P, ImageBuffers B)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [TP][KP+1]
    __shared__ double red[4];
    int n, wg;
    if (!image_wg(P.nimg, P.parts, (P.N + TP - 1) / TP, n, wg)) return;
    const int p0 = wg * TP;
    const int KP = P.KP, ld = KP + 1;
    const size_t hw = (size_t)P.N;
    const float *val = (final_parity ? B.val1 : B.val0) + (size_t)n * ((size_t)P.Mmax + 1) * KP;
    const int *off = B.offset + ((size_t)n * P.Npad + p0) * PD1;
    const float *bar = B.bary + ((size_t)n * P.Npad + p0) * PD1;
    const float alpha = 1.0f / (1.0f + 1.0f / (float)(1 << PD));   // 1/(1+2^-d)
    // the tile's vertex rows and weights, once, through LDS: the KP threads of a pixel would each load the same 2 (d + 1) words
    __shared__ int off_s[TP * PD1];
    __shared__ float bar_s[TP * PD1];
    for (int e = threadIdx.x; e < TP * PD1; e += 256) {
        const bool in = p0 + e / PD1 < P.N;
        off_s[e] = in ? off[e] + 1 : 0;
        bar_s[e] = in ? bar[e] * alpha : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TP * KP; e += 256) {
        const int pl = e / KP, k = e - pl * KP;
        float acc = 0.0f;
        if (p0 + pl < P.N && k < K) {
#pragma unroll
            for (int r = 0; r < PD1; r++) acc = acc + bar_s[pl * PD1 + r] * val[(size_t)off_s[pl * PD1 + r] * KP + k];
        }
        tile[pl * ld + k] = acc;
    }
    // gate of every pixel of the tile, once (it is the same for all K channels: max over the channels -> K loads per pixel, not per element)
    __shared__ float gate_s[TP], roi_s[TP], mx_s[4][TP];
    if (seg) {                                              // 4 threads per pixel share the channel loop (max is order-independent)
        const int pl = threadIdx.x & (TP - 1), q4 = threadIdx.x >> 6;
        const int p = p0 + pl;
        float mx = -3.402823466e38f;
        if (p < P.N) {
            const float *sg = seg + (size_t)n * K * hw + p;
            for (int kk = q4; kk < K; kk += 4) { const float t = sg[(size_t)kk * hw]; mx = t > mx ? t : mx; }
        }
        mx_s[q4][pl] = mx;
    }
    __syncthreads();
    if (seg && threadIdx.x < TP) {
        const int p = p0 + threadIdx.x;
        float g = 0.0f, ro = 0.0f;
        if (p < P.N) {
            float mx = mx_s[0][threadIdx.x];
#pragma unroll
            for (int i = 1; i < 4; i++) mx = mx_s[i][threadIdx.x] > mx ? mx_s[i][threadIdx.x] : mx;
            ro = roi[(size_t)n * hw + p];
            g = ro - mx;
            if (unlabel[(size_t)n * hw + p]) g = 1.0f;
            if (g < 0.0f) g = 0.0f;
        }
        gate_s[threadIdx.x] = g;
        roi_s[threadIdx.x] = ro;
    }
    __syncthreads();
    float *out = outs + (size_t)n * K * hw;
    double part = 0.0;
    for (int e = threadIdx.x; e < TP * K; e += 256) {
        const int k = e / TP, pl = e - k * TP;
        const int p = p0 + pl;
        if (p >= P.N) continue;
        float v = tile[pl * ld + k];
        if (seg) {
            v = v * gate_s[pl];
            part += (double)(seg[(size_t)n * K * hw + (size_t)k * hw + p] * roi_s[pl]) * (double)v;
        }
        out[(size_t)k * hw + p] = v;
    }
    if (seg) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) B.loss_part[(size_t)n * ((P.N + TP - 1) / TP) + wg] = red[0] + red[1] + red[2] + red[3];
    }
}

// loss = -(sum of the workgroups' partial sums) / N, added up in a fixed order (no atomics: the same bits every run)
__global__ __launch_bounds__(256) void energy_finalize_kernel(const double *__restrict__ part, int nparts, float *__restrict__ loss, int N)
{
    __shared__ double red[256];
    double t = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) t += part[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(-red[0] / (double)N);
}

__global__ __launch_bounds__(256) void energy_backward_kernel(const float *__restrict__ AS, const float *__restrict__ roi,
                                                             const float *__restrict__ gout, float *__restrict__ gseg,
                                                             int N, int K, size_t hw)
{
    const size_t tot = (size_t)N * K * hw;
    const float g = gout[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const size_t n = i / ((size_t)K * hw);
        const size_t p = i % hw;
        // -2 * grad_output * AS / N, then * ROI   (utils/seg_helper.py:899-902)
        float v = -2.0f * g;
        v = v * AS[i];
        v = v / (float)N;
        gseg[i] = v * roi[n * hw + p];
    }
}

struct Plan {
    LatticeParams P;
    ImageBuffers B;
    size_t bytes;
};

inline int round_kp(int K) { return (K + 3) & ~3; }

// XCDs per image: the smallest of 1 / 2 / 4 / 8 that keeps >= 80 % of the XCDs busy over the rounds of units (8 always does)
inline int image_parts(int N)
{
    for (int parts = 1; parts < 8; parts *= 2) {
        const int units = N * parts, rounds = (units + 7) / 8;
        if (units * 5 >= rounds * 8 * 4) return parts;
    }
    return 8;
}

inline unsigned image_grid(const LatticeParams &P, int per_image, int parts = 0)
{
    if (parts == 0) parts = P.parts;
    const int lp = (per_image + parts - 1) / parts, units = P.nimg * parts;
    return (unsigned)(8 * ((units + 7) / 8) * lp);
}

size_t plan_layout(int N, int K, int H, int W, void *ws, Plan *pl)
{
    LatticeParams &P = pl->P;
    P.H = H; P.W = W; P.N = H * W; P.Npad = (P.N + 3) & ~3;
    P.nimg = N; P.parts = image_parts(N);
    // worst case (every pixel six points of its own) loads the table to 2/3; the training images fill a few per cent of it
    size_t cap = 1;
    while (cap * 2 < (size_t)P.Npad * PD1 * 3) cap <<= 1;
    P.cap_mask = (unsigned)(cap - 1);
    P.Mmax = P.Npad * PD1;
    P.KP = round_kp(K);
    Carver cv(ws);
    ImageBuffers &B = pl->B;
    B.err = cv.take<int>(64);
    B.M = cv.take<int>((size_t)N);
    B.loss_acc = cv.take<double>(8);
    const size_t head = cv.off;       // [err | M | loss_acc] zeroed every call
    B.table = cv.take<TableEntry>((size_t)N * cap);
    B.pkey = cv.take<unsigned long long>((size_t)N * P.Mmax);
    B.offset = cv.take<int>((size_t)N * P.Npad * PD1);
    B.bary = cv.take<float>((size_t)N * P.Npad * PD1);
    B.nb = cv.take<int2>((size_t)N * PD1 * P.Mmax);
    B.val0 = cv.take<float>((size_t)N * ((size_t)P.Mmax + 1) * P.KP);
    B.val1 = cv.take<float>((size_t)N * ((size_t)P.Mmax + 1) * P.KP);
    const size_t pairs = (size_t)N * P.Npad * PD1;
    B.ckey0 = cv.take<unsigned>(pairs);
    B.ckey1 = cv.take<unsigned>(pairs);
    B.cent0 = cv.take<unsigned>(pairs);
    B.cent1 = cv.take<unsigned>(pairs);
    B.seg_lo = cv.take<int>((size_t)N * P.Mmax);
    B.seg_hi = cv.take<int>((size_t)N * P.Mmax);
    B.rows = cv.take<float>((size_t)N * P.N * P.KP);
    B.loss_part = cv.take<double>((size_t)N * ((P.N + TP - 1) / TP));
    B.id_bits = 1;
    while ((1u << B.id_bits) <= (unsigned)P.Mmax) B.id_bits++;          // ids 0 .. Mmax (Mmax = "padding pixel")
    B.sort_tmp_bytes = rs_scratch_bytes(pairs);
    B.sort_tmp = cv.take<char>(B.sort_tmp_bytes);
    (void)head;
    pl->bytes = cv.off;
    return cv.off;
}

// Phase A -- everything that depends on the image only: hash table, lattice points, per-pixel offsets and barycentric weights,
// blur neighbours.  Phase B -- splat / blur / slice of one value tensor through that lattice.  The training step builds the lattice
// of its strong image on a side stream while the networks run (cosa_dense_energy_prepare) and only phase B waits for the logits.
int setup_plan(int N, int K, int H, int W, float sigmargb, float sigmaxy, void *ws, size_t ws_bytes, Plan &pl)
{
    COSA_REQUIRE(ws, "bilateral: null workspace");
    COSA_REQUIRE(N > 0 && K > 0 && H > 0 && W > 0 && N <= 65535, "bilateral: bad shape");
    COSA_REQUIRE(sigmargb > 0.f && sigmaxy > 0.f, "bilateral: sigmas must be positive");
    COSA_REQUIRE((size_t)H * W * PD1 < (1u << 30), "bilateral: image too large");
    {
        int idb = 1, nb = 0;
        while ((1u << idb) <= (unsigned)(((size_t)H * W + 3) / 4 * 4 * PD1)) idb++;
        while ((1 << nb) < N) nb++;
        COSA_REQUIRE(idb + nb <= 32, "bilateral: batch x lattice size does not fit the 32-bit sort key (split the batch)");
    }
    if (ws_bytes < plan_layout(N, K, H, W, ws, &pl)) {
        set_error("bilateral: workspace too small (%zu < %zu)", ws_bytes, pl.bytes);
        return COSA_ENOMEM;
    }
    LatticeParams &P = pl.P;
    P.sigmaxy = sigmaxy; P.sigmargb = sigmargb;
    // permutohedral.cpp:152-156: float inv_std_dev = sqrt(2/3)*(d+1); scale = 1/sqrt((i+2)(i+1)) * inv_std_dev (double)
    const float inv_std_dev = (float)(std::sqrt(2.0 / 3.0) * (PD + 1));
    for (int i = 0; i < PD; i++) P.scale[i] = (float)(1.0 / std::sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);
    return COSA_OK;
}

int lattice_phase(const float *images, int N, Plan &pl, hipStream_t st)
{
    COSA_REQUIRE(images || PD == 2, "bilateral: null image pointer");
    LatticeParams &P = pl.P;
    ImageBuffers &B = pl.B;
    const size_t cap = (size_t)P.cap_mask + 1;
    // zero [err | M | loss_acc] (one block at the start of the workspace), fill the table with EMPTY keys (ids: don't care)
    COSA_HIP_CHECK(hipMemsetAsync(B.err, 0, (char *)B.table - (char *)B.err, st));
    COSA_HIP_CHECK(hipMemsetAsync(B.table, 0xFF, (size_t)N * cap * sizeof(TableEntry), st));
    const dim3 blk(256);
    hipLaunchKernelGGL(lattice_build_kernel, dim3(image_grid(P, (P.Npad + 255) / 256)), blk, 0, st, images, P, B);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(lattice_remap_kernel, dim3(image_grid(P, (P.Npad * PD1 + 255) / 256)), blk, 0, st, P, B);
    COSA_LAUNCH_CHECK();
    const int gs = kWgPerImage;   // grid-stride launches read M on the device
    hipLaunchKernelGGL(lattice_neighbors_kernel, dim3(image_grid(P, gs)), blk, 0, st, gs, P, B);
    COSA_LAUNCH_CHECK();
    // the splat lists: pairs sorted by (image, vertex); stable, so a vertex keeps its pairs in pixel order
    const size_t pairs = (size_t)N * P.Npad * PD1;
    int nbits = 0;
    while ((1 << nbits) < N) nbits++;
    const int where = rs_sort_pairs(B.ckey0, B.cent0, B.ckey1, B.cent1, pairs, B.id_bits + nbits, B.sort_tmp, st);
    COSA_REQUIRE(where >= 0, "bilateral: radix sort launch failed");
    if (where == 0) {       // an even number of 8-bit passes (small images): the consumers read buffer 1
        COSA_HIP_CHECK(hipMemcpyAsync(B.ckey1, B.ckey0, pairs * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
        COSA_HIP_CHECK(hipMemcpyAsync(B.cent1, B.cent0, pairs * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
    }
    COSA_HIP_CHECK(hipMemsetAsync(B.seg_lo, 0, (char *)B.rows - (char *)B.seg_lo, st));        // seg_lo and seg_hi are adjacent
    hipLaunchKernelGGL(lattice_bounds_kernel, dim3((unsigned)((pairs + 255) / 256)), blk, 0, st, pairs, P, B);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

int filter_phase(const float *ins, float *outs, int N, int K, const float *roi, const float *seg_for_energy,
                 const unsigned char *unlabel, float *loss, int32_t *lattice_sizes, Plan &pl, hipStream_t st)
{
    COSA_REQUIRE(ins && outs, "bilateral: null pointer");
    LatticeParams &P = pl.P;
    ImageBuffers &B = pl.B;
    const dim3 blk(256);
    const int gs = kWgPerImage;
    const int tiles = (P.N + TP - 1) / TP;
    const size_t lds = (size_t)TP * (P.KP + 1) * sizeof(float);
    hipLaunchKernelGGL(lattice_rows_kernel, dim3(image_grid(P, tiles, kAllXcds)), blk, lds, st, ins, roi, K, P, B);
    COSA_LAUNCH_CHECK();
    if (P.KP <= 32) hipLaunchKernelGGL(lattice_splat_sorted_kernel<1>, dim3(image_grid(P, gs)), blk, 0, st, gs, P, B);
    else if (P.KP <= 64) hipLaunchKernelGGL(lattice_splat_sorted_kernel<2>, dim3(image_grid(P, gs)), blk, 0, st, gs, P, B);
    else hipLaunchKernelGGL(lattice_splat_sorted_kernel<3>, dim3(image_grid(P, gs)), blk, 0, st, gs, P, B);
    COSA_LAUNCH_CHECK();
    for (int j = 0; j <= PD; j++) {
        hipLaunchKernelGGL(lattice_blur_kernel, dim3(image_grid(P, gs, kAllXcds)), blk, 0, st, j, j & 1, gs, P, B);
        COSA_LAUNCH_CHECK();
    }
    // d + 1 ping-pong passes: 6 (5-D lattice) leave the result in val0, 3 (2-D lattice) in val1
    hipLaunchKernelGGL(lattice_slice_kernel, dim3(image_grid(P, tiles)), blk, lds, st, outs, K, PD1 & 1, seg_for_energy, roi,
                       unlabel, P, B);
    COSA_LAUNCH_CHECK();
    if (seg_for_energy) {
        hipLaunchKernelGGL(energy_finalize_kernel, dim3(1), dim3(256), 0, st, B.loss_part, N * ((P.N + TP - 1) / TP), loss, N);
        COSA_LAUNCH_CHECK();
    }
    if (lattice_sizes) COSA_HIP_CHECK(hipMemcpyAsync(lattice_sizes, B.M, sizeof(int) * N, hipMemcpyDeviceToDevice, st));
    return COSA_OK;
}

int run_filter(const float *images, const float *ins, float *outs, int N, int K, int H, int W, float sigmargb, float sigmaxy,
               const float *roi, const float *seg_for_energy, const unsigned char *unlabel, float *loss,
               int32_t *lattice_sizes, void *ws, size_t ws_bytes, hipStream_t st)
{
    Plan pl;
    int rc = setup_plan(N, K, H, W, sigmargb, sigmaxy, ws, ws_bytes, pl);
    if (rc) return rc;
    rc = lattice_phase(images, N, pl, st);
    if (rc) return rc;
    return filter_phase(ins, outs, N, K, roi, seg_for_energy, unlabel, loss, lattice_sizes, pl, st);
}

// half-resolution, de-normalised image of the dense-energy regulariser: F.interpolate(denormalize_img(x), scale_factor=0.5)
// (nearest: pixel (2y, 2x)); x*std + mean as two operations, like the reference's tensor expression
__global__ __launch_bounds__(256) void half_denorm_kernel(const float *__restrict__ simg, float *__restrict__ out, int S, int planes)
{
    const int Sq = S >> 1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)planes * Sq * Sq) return;
    const int x = (int)(i % Sq), y = (int)((i / Sq) % Sq), pl = (int)(i / ((size_t)Sq * Sq));
    const int ch = pl % 3;
    const float mean = ch == 0 ? 123.675f : (ch == 1 ? 116.28f : 103.53f), sd = ch == 0 ? 58.395f : (ch == 1 ? 57.12f : 57.375f);
    const float v = simg[((size_t)pl * S + 2 * y) * S + 2 * x] * sd;
    out[i] = v + mean;
}

}  // namespace
}  // namespace cosa

using namespace cosa;

#if COSA_PD == 2
// ---- position-only Gaussian kernel (2-D lattice) of the dense-CRF post-processing ------------------------------------------------------
extern "C" size_t cosa_lattice_filter_d2_workspace_bytes(int N, int K, int H, int W)
{
    if (N <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    Plan pl;
    return plan_layout(N, K, H, W, nullptr, &pl);
}

extern "C" int cosa_lattice_filter_d2(const float *ins, float *outs, int N, int K, int H, int W, float sigmaxy, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    return run_filter(nullptr, ins, outs, N, K, H, W, 1.0f, sigmaxy, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                      as_stream(stream));
}
#else
// the lattice's stable LSD radix sort on its own (csrc/radix_sort.hpp), for the test-suite: n pairs sorted by the low `bits` bits of the key into
// (keys_out, vals_out); keys_in / vals_in are scratch afterwards; workspace >= cosa_radix_sort_workspace_bytes(n)
extern "C" size_t cosa_radix_sort_workspace_bytes(long long n) { return n > 0 ? rs_scratch_bytes((size_t)n) : 0; }
extern "C" int cosa_radix_sort_pairs(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, long long n, int bits,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(keys_in && vals_in && keys_out && vals_out && workspace && n > 0 && n < (1ll << 32) && bits > 0 && bits <= 32, "cosa_radix_sort_pairs: bad arguments");
    COSA_REQUIRE(workspace_bytes >= rs_scratch_bytes((size_t)n), "cosa_radix_sort_pairs: workspace too small");
    hipStream_t st = as_stream(stream);
    const int where = rs_sort_pairs(keys_in, vals_in, keys_out, vals_out, (size_t)n, bits, workspace, st);
    COSA_REQUIRE(where >= 0, "cosa_radix_sort_pairs: launch failed");
    if (where == 0) {
        COSA_HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        COSA_HIP_CHECK(hipMemcpyAsync(vals_out, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    }
    return COSA_OK;
}

extern "C" size_t cosa_bilateral_workspace_bytes(int N, int K, int H, int W)
{
    if (N <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    Plan pl;
    return plan_layout(N, K, H, W, nullptr, &pl);
}

extern "C" int cosa_bilateralfilter_batch_dev(const float *images, const float *ins, float *outs,
                                              int N, int K, int H, int W, float sigmargb, float sigmaxy,
                                              int32_t *lattice_sizes, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_filter(images, ins, outs, N, K, H, W, sigmargb, sigmaxy, nullptr, nullptr, nullptr, nullptr, lattice_sizes,
                      workspace, workspace_bytes, as_stream(stream));
}

extern "C" int cosa_dense_energy_forward(const float *images, const float *seg, const float *roi, const uint8_t *unlabel,
                                         float *AS, float *loss, int N, int K, int H, int W, float sigmargb, float sigmaxy,
                                         void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(seg && roi && unlabel && AS && loss, "cosa_dense_energy_forward: null pointer");
    return run_filter(images, seg, AS, N, K, H, W, sigmargb, sigmaxy, roi, seg, unlabel, loss, nullptr, workspace,
                      workspace_bytes, as_stream(stream));
}

// The same forward, in two halves for overlap: `prepare` needs only the (normalised) strong image -- it writes the half-resolution
// de-normalised image and builds the lattice in `workspace`; `forward_prepared` runs splat / blur / slice through that lattice.
// The K passed to both (and to cosa_bilateral_workspace_bytes) must agree; the workspace must not be touched in between.
extern "C" int cosa_dense_energy_prepare(const float *simg, float *s_img, int N, int K, int S, float sigmargb, float sigmaxy,
                                         void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(simg && s_img && S > 0 && (S % 2) == 0, "cosa_dense_energy_prepare: bad arguments");
    Plan pl;
    int rc = setup_plan(N, K, S / 2, S / 2, sigmargb, sigmaxy, workspace, workspace_bytes, pl);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const size_t tot = (size_t)N * 3 * (S / 2) * (S / 2);
    hipLaunchKernelGGL(half_denorm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, simg, s_img, S, N * 3);
    COSA_LAUNCH_CHECK();
    return lattice_phase(s_img, N, pl, st);
}

extern "C" int cosa_dense_energy_forward_prepared(const float *seg, const float *roi, const uint8_t *unlabel, float *AS, float *loss,
                                                  int N, int K, int H, int W, float sigmargb, float sigmaxy, void *workspace,
                                                  size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(seg && roi && unlabel && AS && loss, "cosa_dense_energy_forward_prepared: null pointer");
    Plan pl;
    int rc = setup_plan(N, K, H, W, sigmargb, sigmaxy, workspace, workspace_bytes, pl);
    if (rc) return rc;
    return filter_phase(seg, AS, N, K, roi, seg, unlabel, loss, nullptr, pl, as_stream(stream));
}

extern "C" int cosa_dense_energy_backward(const float *AS, const float *roi, const float *grad_out, float *grad_seg,
                                          int N, int K, int H, int W, void *stream)
{
    COSA_REQUIRE(AS && roi && grad_out && grad_seg && N > 0 && K > 0 && H > 0 && W > 0, "cosa_dense_energy_backward: bad arguments");
    const size_t hw = (size_t)H * W;
    const size_t tot = (size_t)N * K * hw;
    int grid = (int)((tot + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(energy_backward_kernel, dim3(grid), dim3(256), 0, as_stream(stream), AS, roi, grad_out, grad_seg, N, K, hw);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// ---- host-pointer drop-ins for the SWIG module (bilateralfilter.hpp:10-12) -----------------------------
static int host_filter(float *images, float *ins, float *outs, int N, int K, int H, int W, float srgb, float sxy)
{
    const size_t hw = (size_t)H * W;
    const size_t ws_bytes = cosa_bilateral_workspace_bytes(N, K, H, W);
    float *d_img = nullptr, *d_in = nullptr, *d_out = nullptr;
    void *d_ws = nullptr;
    int rc = COSA_OK, err = 0;
    hipStream_t st = nullptr;
#define HF_CHECK(e) do { if ((e) != hipSuccess) { set_error("bilateralfilter: %s failed", #e); rc = COSA_EHIP; goto done; } } while (0)
    HF_CHECK(hipMalloc(&d_img, (size_t)N * 3 * hw * sizeof(float)));
    HF_CHECK(hipMalloc(&d_in, (size_t)N * K * hw * sizeof(float)));
    HF_CHECK(hipMalloc(&d_out, (size_t)N * K * hw * sizeof(float)));
    HF_CHECK(hipMalloc(&d_ws, ws_bytes));
    HF_CHECK(hipMemcpy(d_img, images, (size_t)N * 3 * hw * sizeof(float), hipMemcpyHostToDevice));
    HF_CHECK(hipMemcpy(d_in, ins, (size_t)N * K * hw * sizeof(float), hipMemcpyHostToDevice));
    rc = run_filter(d_img, d_in, d_out, N, K, H, W, srgb, sxy, nullptr, nullptr, nullptr, nullptr, nullptr, d_ws, ws_bytes, st);
    if (rc) goto done;
    HF_CHECK(hipMemcpy(outs, d_out, (size_t)N * K * hw * sizeof(float), hipMemcpyDeviceToHost));
    HF_CHECK(hipMemcpy(&err, d_ws, sizeof(int), hipMemcpyDeviceToHost));
    if (err) { set_error("bilateralfilter: lattice key out of packable range (sigma too small)"); rc = COSA_ERANGE; }
done:
#undef HF_CHECK
    if (d_img) (void)hipFree(d_img);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

extern "C" void bilateralfilter(float *image, int len_image, float *in, int len_in, float *out, int len_out,
                                int H, int W, float sigmargb, float sigmaxy)
{
    (void)len_image; (void)len_out;
    const int K = len_in / W / H;   // bilateralfilter.cpp:27
    if (host_filter(image, in, out, 1, K, H, W, sigmargb, sigmaxy) != COSA_OK)
        fprintf(stderr, "cosa bilateralfilter: %s\n", cosa_last_error());
}

extern "C" void bilateralfilter_batch(float *images, int len_images, float *ins, int len_ins, float *outs, int len_outs,
                                      int N, int K, int H, int W, float sigmargb, float sigmaxy)
{
    (void)len_images; (void)len_ins; (void)len_outs;
    if (host_filter(images, ins, outs, N, K, H, W, sigmargb, sigmaxy) != COSA_OK)
        fprintf(stderr, "cosa bilateralfilter_batch: %s\n", cosa_last_error());
}
#endif
