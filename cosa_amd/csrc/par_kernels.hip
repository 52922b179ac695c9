// par_kernels.hip -- Pixel-Adaptive Refinement (models/PAR.py:26-91) for gfx950.
//
// Compiled with -ffp-contract=off; follows DESIGN.md "Arithmetic spec" (spec P) so that the
// refined masks -- and the label maps cut from them -- are bit-identical to the CPU oracle.
//
//   affinity:  per pixel, 8 neighbours x n_dil dilations (replicate border):
//              aff_n = softmax_n( mean_c( -(|x_n - x| / (std_n(x) + 1e-8) / 0.3)^2 ) ) + 0.01 * posw_n
//   step:      m'[k] = sum_n aff_n * m[k](neighbour n)        (num_iter times, ping-pong)
//
// Layout: aff [B][NN][h*w] (plane per neighbour: coalesced across the wavefront),
//         masks [B][planes][h*w].
#include <cstdlib>
#include "kernels.hpp"
#include <cmath>

namespace cosa {
namespace {

__device__ __forceinline__ float cosa_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(k, -0.693359375f, x);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float e = __builtin_fmaf(p, r2, r);
    e = e + 1.0f;
    return __builtin_ldexpf(e, (int)k);
}

// host twin of the same spec (for the constant position prior)
inline float host_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = std::rint(x * 1.44269504088896341f);
    float r = std::fma(k, -0.693359375f, x);
    r = std::fma(k, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = std::fma(p, r, 1.3981999507E-3f);
    p = std::fma(p, r, 8.3334519073E-3f);
    p = std::fma(p, r, 4.1665795894E-2f);
    p = std::fma(p, r, 1.6666665459E-1f);
    p = std::fma(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float e = std::fma(p, r2, r);
    e = e + 1.0f;
    return std::ldexp(e, (int)k);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// neighbour t of get_kernel (models/PAR.py:10-24): (dy,dx) in {-1,0,1}
__device__ __forceinline__ void nbr_off(int t, int &dy, int &dx)
{
    // t: 0 1 2 3 4 5 6 7 -> dy: -1 -1 -1 0 0 1 1 1 ; dx: -1 0 1 -1 1 -1 0 1
    const int tt = t < 4 ? t : t + 1;   // skip the centre of the 3x3
    dy = tt / 3 - 1;
    dx = tt % 3 - 1;
}

__global__ __launch_bounds__(256) void par_affinity_kernel(const float *__restrict__ imgs, float *__restrict__ aff,
                                                          int h, int w, ParPlan plan)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int hw = h * w;
    if (pix >= hw) return;
    const int b = blockIdx.y;
    const int y = pix / w, x = pix - y * w;
    const int NN = plan.n_dil * 8;
    const float *img = imgs + (size_t)b * 3 * hw;
    float sd[3], ctr[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float *pl = img + (size_t)c * hw;
        ctr[c] = pl[pix];
        float sum = 0.0f;
        for (int n = 0; n < NN; n++) {
            int dy, dx;
            nbr_off(n & 7, dy, dx);
            const int d = plan.dil[n >> 3];
            sum = sum + pl[clampi(y + dy * d, 0, h - 1) * w + clampi(x + dx * d, 0, w - 1)];
        }
        const float mean = sum / (float)NN;
        float var = 0.0f;
        for (int n = 0; n < NN; n++) {
            int dy, dx;
            nbr_off(n & 7, dy, dx);
            const int d = plan.dil[n >> 3];
            float dl = pl[clampi(y + dy * d, 0, h - 1) * w + clampi(x + dx * d, 0, w - 1)] - mean;
            var = var + dl * dl;
        }
        var = var / (float)(NN - 1);
        sd[c] = __builtin_sqrtf(var) + 1e-8f;
    }
    float *out = aff + (size_t)b * NN * hw + pix;
    // logits are staged through the output buffer (own slot per thread): pass 1 logits + max
    float mx = -INFINITY;
    for (int n = 0; n < NN; n++) {
        int dy, dx;
        nbr_off(n & 7, dy, dx);
        const int d = plan.dil[n >> 3];
        const int o = clampi(y + dy * d, 0, h - 1) * w + clampi(x + dx * d, 0, w - 1);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float a = __builtin_fabsf(img[(size_t)c * hw + o] - ctr[c]);
            float q = a / sd[c];
            q = q / 0.3f;
            acc = acc + (-(q * q));
        }
        const float lg = acc / 3.0f;
        out[(size_t)n * hw] = lg;
        mx = lg > mx ? lg : mx;
    }
    float es = 0.0f;
    for (int n = 0; n < NN; n++) {
        const float e = cosa_expf(out[(size_t)n * hw] - mx);
        out[(size_t)n * hw] = e;
        es = es + e;
    }
    for (int n = 0; n < NN; n++) {
        const float a = out[(size_t)n * hw] / es;
        out[(size_t)n * hw] = a + 0.01f * plan.posw[n];
    }
}

// IEEE-754 binary32 division with the divisor's part hoisted.  `x / d` on gfx950 is lowered to: (scale), r0 = rcp(d), e = fma(-d, r0, 1),
// r = fma(e, r0, r0), q0 = x * r, e1 = fma(-d, q0, x), q1 = fma(e1, r, q0), e2 = fma(-d, q1, x), q = fma(e2, r, q1), (fixup): the
// correctly rounded quotient.  The scale / fixup steps only act on operands near the ends of the exponent range; here 0 <= x <= 1e9
// and 1e-8 <= d <= 1e9, so they are identities and the SAME sequence with r computed once per divisor gives the same bits for a
// third of the instructions (the affinity kernel divides 48 values by each sigma, by 0.3 and by the softmax sum: 390 divisions / pixel).
struct ExactRcp { float d, r; };
__device__ __forceinline__ ExactRcp exact_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return ExactRcp{d, __builtin_fmaf(e, r0, r0)};
}
__device__ __forceinline__ float exact_div(float x, const ExactRcp &k)
{
    const float q0 = x * k.r;
    const float e1 = __builtin_fmaf(-k.d, q0, x);
    const float q1 = __builtin_fmaf(e1, k.r, q0);
    const float e2 = __builtin_fmaf(-k.d, q1, x);
    return __builtin_fmaf(e2, k.r, q1);
}

// affinity of the named configuration: the three image channels of a 16 x 16 pixel tile plus its 24-pixel replicate-clamped halo staged in LDS once (48 KB), every
// tap an LDS read at a compile-time offset.  The taps of a channel are read three times (sum, variance, logits) instead of being held in 48
// registers: ~80 VGPRs, six waves per SIMD, no global-load latency inside the arithmetic.  Same operations, same order as the spec.
struct Dil6a { static constexpr int n = 6; static constexpr int d[6] = {1, 2, 4, 8, 12, 24}; };
constexpr int kATile = 16, kAHalo = 24, kALW = kATile + 2 * kAHalo;        // 64 x 64 floats per channel

template <typename DIL>
__global__ __launch_bounds__(256) void par_affinity_tiled_kernel(const float *__restrict__ imgs, float *__restrict__ aff,
                                                             int h, int w, ParPlan plan, int tiles_x)
{
    __shared__ float timg[3][kALW * kALW];
    constexpr int ND = DIL::n, NN = ND * 8;
    static_assert(DIL::d[ND - 1] <= kAHalo, "every dilation must fit the halo");
    const int b = blockIdx.y;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int y0 = tyi * kATile, x0 = txi * kATile;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int hw = h * w;
    const float *img = imgs + (size_t)b * 3 * hw;
    // staging: 4096 elements per channel, 16 per thread; consecutive threads read consecutive columns of a row
#pragma unroll
    for (int i = 0; i < (kALW * kALW) / 256; i++) {
        const int e = tid + 256 * i;
        const int r = e / kALW, c = e - r * kALW;
        const int go = clampi(y0 - kAHalo + r, 0, h - 1) * w + clampi(x0 - kAHalo + c, 0, w - 1);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) timg[ch][e] = img[(size_t)ch * hw + go];
    }
    __syncthreads();
    const int y = y0 + ty, x = x0 + tx;
    if (y >= h || x >= w) return;
    const int pix = y * w + x;
    const int ctr = (ty + kAHalo) * kALW + tx + kAHalo;
    const ExactRcp k03 = exact_rcp(0.3f), k3 = exact_rcp(3.0f);
    float lg[NN];
#pragma unroll
    for (int n = 0; n < NN; n++) lg[n] = 0.0f;
#define COSA_AFF_TAP(ch, n) timg[ch][ctr + ((((n) & 7) < 4 ? ((n) & 7) : ((n) & 7) + 1) / 3 - 1) * DIL::d[(n) >> 3] * kALW + ((((n) & 7) < 4 ? ((n) & 7) : ((n) & 7) + 1) % 3 - 1) * DIL::d[(n) >> 3]]
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        const float cv = timg[c][ctr];
        float sum = 0.0f;
#pragma unroll
        for (int n = 0; n < NN; n++) sum = sum + COSA_AFF_TAP(c, n);
        const float mean = sum / (float)NN;
        float var = 0.0f;
#pragma unroll
        for (int n = 0; n < NN; n++) { const float dl = COSA_AFF_TAP(c, n) - mean; var = var + dl * dl; }
        var = var / (float)(NN - 1);
        const ExactRcp ksd = exact_rcp(__builtin_sqrtf(var) + 1e-8f);
#pragma unroll
        for (int n = 0; n < NN; n++) {
            float q = exact_div(__builtin_fabsf(COSA_AFF_TAP(c, n) - cv), ksd);
            q = exact_div(q, k03);
            lg[n] = lg[n] + (-(q * q));
        }
    }
#undef COSA_AFF_TAP
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < NN; n++) { lg[n] = -exact_div(-lg[n], k3); mx = lg[n] > mx ? lg[n] : mx; }
    float es = 0.0f;
#pragma unroll
    for (int n = 0; n < NN; n++) { lg[n] = cosa_expf(lg[n] - mx); es = es + lg[n]; }
    const ExactRcp kes = exact_rcp(es);
    float *out = aff + (size_t)b * NN * hw + pix;
#pragma unroll
    for (int n = 0; n < NN; n++) out[(size_t)n * hw] = exact_div(lg[n], kes) + 0.01f * plan.posw[n];
}


// generic propagation step (any dilation list): ONE pixel per thread, ALL live planes of the image (up to 16) in the same thread.  The
// affinity tensor is the HBM stream of a step (rocprofv3 PMC, profiles/r01_par_step_pmc.json: 595 MB per step launch when planes were
// walked in groups of 4 = the 154 MB tensor once per group); with every plane in one thread it is read once.  One pixel per thread keeps
// the registers small (16 accumulators + 16 plane bases).
template <int NP>
__device__ __forceinline__ void par_step_body(const float *__restrict__ aff, const float *__restrict__ src, float *__restrict__ dst,
                                               int b, int pblk, int K, int j0, int nlive, int half_planes, size_t img_stride, int h,
                                               int w, const ParPlan &plan)
{
    const int pix = pblk * 256 + threadIdx.x;
    const int hw = h * w;
    if (pix >= hw) return;
    const int y = pix / w, x = pix - y * w;
    unsigned po[NP];                                   // plane offsets (floats) from the image base
    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int j = j0 + (i < nlive ? i : 0);
        const int half = j / K;
        po[i] = (unsigned)((half * half_planes + (j - half * K)) * hw);
        acc[i] = 0.0f;
    }
    const float *sb = src + (size_t)b * img_stride;
    const float *ab = aff + (size_t)b * plan.n_dil * 8 * hw + pix;
    for (int di = 0; di < plan.n_dil; di++) {
        const int d = plan.dil[di];
        const int rm = clampi(y - d, 0, h - 1) * w, r0 = y * w, rp = clampi(y + d, 0, h - 1) * w;
        const int xm = clampi(x - d, 0, w - 1), xp = clampi(x + d, 0, w - 1);
        const int o[8] = {rm + xm, rm + x, rm + xp, r0 + xm, r0 + xp, rp + xm, rp + x, rp + xp};
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const float a = ab[(size_t)(di * 8 + t) * hw];
#pragma unroll
            for (int i = 0; i < NP; i++) acc[i] = acc[i] + sb[po[i] + o[t]] * a;
        }
    }
#pragma unroll
    for (int i = 0; i < NP; i++)
        if (i < nlive) dst[(size_t)b * img_stride + po[i] + pix] = acc[i];
}

// Workgroup ids are dealt round-robin to the 8 XCDs (id & 7), each with its own L2.  The taps of a pixel reach +-24 rows, so with
// pixel blocks of one image spread over all XCDs every L2 ends up fetching every mask plane (PMC: 506 MB of fetches per launch for
// 154 MB of affinities + 38 MB of masks).  Images are therefore pinned to XCDs: id -> (xcd = id & 7, j = id >> 3), image =
// 8 * (j / blocks_per_image) + xcd, so an L2 only ever sees the planes of "its" images.
__global__ __launch_bounds__(256) void par_step_kernel(const float *__restrict__ aff, const float *__restrict__ src,
                                                       float *__restrict__ dst, const int *__restrict__ kcount, int Kfull,
                                                       int halves, int half_planes, size_t img_stride, int h, int w, ParPlan plan,
                                                       int B, int pix_blocks, int groups, int pin)
{
    const int id = blockIdx.x;
    const int per_img = pix_blocks * groups;
    int b, rem;
    if (pin) {
        const int xcd = id & 7, j = id >> 3, slot = j / per_img;
        rem = j - slot * per_img;
        b = slot * 8 + xcd;
    } else {                                           // few images: spread every image over all XCDs instead
        b = id / per_img;
        rem = id - b * per_img;
    }
    if (b >= B) return;
    const int grp = rem / pix_blocks, pblk = rem - grp * pix_blocks;
    const int K = kcount ? kcount[b] : Kfull;
    const int live = K * halves;
    const int j0 = grp * 16;
    if (j0 >= live) return;
    int nlive = live - j0;
    nlive = nlive > 16 ? 16 : nlive;
    if (nlive <= 2) par_step_body<2>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
    else if (nlive <= 4) par_step_body<4>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
    else if (nlive <= 6) par_step_body<6>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
    else if (nlive <= 8) par_step_body<8>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
    else if (nlive <= 12) par_step_body<12>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
    else par_step_body<16>(aff, src, dst, b, pblk, K, j0, nlive, half_planes, img_stride, h, w, plan);
}


// propagation step of the named configuration: LDS-tiled, two planes per pass.  A 256-thread workgroup owns a TH x TW = 8 x 32 pixel tile
// of ONE image and walks ALL its live planes in PAIRS:
//   * the 48 affinities of a pixel are loaded ONCE per step into registers and reused for every plane;
//   * per plane pair, the tile plus a HALO-pixel ring (replicate-clamped at the image border, exactly the reference's F.pad) is staged in LDS
//     as (plane j, plane j + 1) float pairs -- 1792 pairs for 256 pixels -- and the taps of every dilation d <= HALO (40 of the 48 for the
//     named configuration 1, 2, 4, 8, 12, 24) are ds_read_b64 at compile-time offsets: one LDS instruction serves both planes at twice the
//     bytes per clock of ds_read_b32 (the single-plane form of this kernel spent 2.0 of its 3.9 us per plane in the LDS pipe), and the
//     multiply / add of the two planes are one packed instruction each (v_pk_mul_f32 with the affinity broadcast, v_pk_add_f32: separate
//     IEEE roundings per half, as the spec's mul-then-sum); only the d > HALO taps (8) remain global gathers, issued first;
//   * the reads are explicit instructions (left to itself the compiler joins neighbours into ds_read2_b64, which moves the same bytes at
//     half the rate) issued four taps ahead of the arithmetic that consumes them; reads, multiplies and adds are volatile asm, which keeps
//     that interleave (the scheduler otherwise issues all 40 reads first and holds 40 products in registers);
//   * two LDS buffers: the next pair's tile is fetched while this one is consumed, one barrier per pair.
// An odd plane count walks its last plane twice (second copy not stored).  Accumulation order is the spec's (acc = acc + m * a, neighbour
// index ascending): bit-identical to the oracle.
// DIL: compile-time dilation list (the only configuration the reference names, models/PAR.py:94); other lists take par_step_kernel.
constexpr int kTH = 8, kTW = 32, kHalo = 12, kStepThreads = kTH * kTW;      // 256 threads, 119 registers per thread: four workgroups per CU
constexpr int kLW = kTW + 2 * kHalo, kLH = kTH + 2 * kHalo;           // 56 x 32
struct Dil6 { static constexpr int n = 6; static constexpr int d[6] = {1, 2, 4, 8, 12, 24}; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_p;

template <typename DIL>
__global__ __launch_bounds__(kStepThreads, 3) void par_step_tiled_kernel(const float *__restrict__ aff, const float *__restrict__ src,
                                                               float *__restrict__ dst, const int *__restrict__ kcount, int Kfull,
                                                               int halves, int half_planes, size_t img_stride, int h, int w,
                                                               int B, int tiles_x, int tiles_y, int pin)
{
    constexpr int NE = (kLH * kLW + kStepThreads - 1) / kStepThreads;        // staged positions per thread (7)
    __shared__ __attribute__((aligned(16))) f32x2 tile[2][NE * kStepThreads];   // kLH x kLW halo tile of plane pairs, padded so that every thread stages NE positions unconditionally
    constexpr int ND = DIL::n, NN = ND * 8;
    const int id = blockIdx.x, per_img = tiles_x * tiles_y;
    int b, rem;
    if (pin) {                                         // images pinned to XCDs (workgroup ids are dealt round-robin to the 8 L2s)
        const int xcd = id & 7, j = id >> 3, slot = j / per_img;
        rem = j - slot * per_img;
        b = slot * 8 + xcd;
    } else {
        b = id / per_img;
        rem = id - b * per_img;
    }
    if (b >= B) return;
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int y0 = tyi * kTH, x0 = txi * kTW;
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int y = y0 + ty, x = x0 + tx;
    const bool valid = y < h && x < w;
    const int hw = h * w;
    const int K = __builtin_amdgcn_readfirstlane(kcount ? kcount[b] : Kfull);     // (uniform: keeps the plane offsets in scalar registers)
    const int live = K * halves;
    if (live <= 0) return;
    const int pix = valid ? y * w + x : 0;

    // affinities n = 2k, 2k + 1 share a register pair: the packed multiply broadcasts either half (op_sel), so the 48 cost 48 registers
    // (written as `m * a` the compiler wants every broadcast operand in the low half of a pair of its own: 96 registers, spills)
    // (global accesses go through buffer descriptors: a 32-bit lane offset + a wave-uniform plane offset per access instead of a 64-bit
    // address computed in two more registers each; the launcher checks that every operand stays below 4 GiB)
    const __amdgpu_buffer_rsrc_t rs_aff = __builtin_amdgcn_make_buffer_rsrc((void *)aff, 0, (unsigned)((size_t)B * NN * hw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (unsigned)((size_t)B * img_stride * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, (unsigned)((size_t)B * img_stride * 4), 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t &rs, unsigned lane_off, unsigned uni_off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, uni_off, 0));
    };
    f32x2 ap[NN / 2];
    {
        const unsigned ab = (unsigned)(b * NN * hw) * 4u;
#pragma unroll
        for (int n = 0; n < NN / 2; n++) ap[n] = (f32x2){ld(rs_aff, pix * 4, ab + (unsigned)(2 * n * hw) * 4u), ld(rs_aff, pix * 4, ab + (unsigned)((2 * n + 1) * hw) * 4u)};
    }
    // acc = acc + (mv[0] * a[n], mv[1] * a[n]), n a compile-time constant after unrolling: one asm statement (multiply, then add: two
    // roundings), volatile so that it stays between the LDS reads it is written between -- left free, the scheduler issues all the reads
    // first and keeps all 40 products in registers
    auto mul_add = [&](f32x2 &acc, const f32x2 &mv, int n) {
        f32x2 p;
        if (n & 1) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(p) : "v"(mv), "v"(ap[n >> 1]));
        else asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(p) : "v"(mv), "v"(ap[n >> 1]));
    };
    // global offsets of the far taps (d > kHalo), clamped; shared by all planes
    unsigned ofar[8];
    {
        constexpr int df = DIL::d[ND - 1];
        const int yc = valid ? y : 0, xc = valid ? x : 0;
        const int rm = clampi(yc - df, 0, h - 1) * w, r0 = yc * w, rp = clampi(yc + df, 0, h - 1) * w;
        const int xm = clampi(xc - df, 0, w - 1), xp = clampi(xc + df, 0, w - 1);
        ofar[0] = rm + xm; ofar[1] = rm + xc; ofar[2] = rm + xp; ofar[3] = r0 + xm; ofar[4] = r0 + xp;
        ofar[5] = rp + xm; ofar[6] = rp + xc; ofar[7] = rp + xp;
#pragma unroll
        for (int t = 0; t < 8; t++) ofar[t] *= 4u;                                     // byte offsets
    }
    static_assert(DIL::d[ND - 1] > kHalo && DIL::d[ND - 2] <= kHalo, "exactly the last dilation is gathered, the others are staged");
    // staging map of this thread: positions e = tid, tid + 256, ... of the (kLH x kLW) halo tile -> clamped image offsets
    unsigned goff[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
        const int e = tid + kStepThreads * i;                   // (positions past the tile land in the padding: any valid address will do)
        const int r = e / kLW, c = e - r * kLW;
        goff[i] = (unsigned)(clampi(y0 - kHalo + r, 0, h - 1) * w + clampi(x0 - kHalo + c, 0, w - 1)) * 4u;        // byte offsets
    }
    auto plane_off = [&](int j) {                      // byte offset of live plane j of this image (wave-uniform)
        j = j < live ? j : live - 1;                   // the partner of an odd count's last plane is that plane again
        const int half = j / K;
        return (unsigned)(((size_t)b * img_stride + (size_t)(half * half_planes + (j - half * K)) * hw) * 4);
    };
    // staging: fetch -> registers, (compute), registers -> LDS, so that all loads of a thread are in flight together
    f32x2 sv[NE];
    auto fetch = [&](int q) {
        const unsigned p0 = plane_off(2 * q), p1 = plane_off(2 * q + 1);
#pragma unroll
        for (int i = 0; i < NE; i++) sv[i] = (f32x2){ld(rs_src, goff[i], p0), ld(rs_src, goff[i], p1)};
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; i++) tile[buf][tid + kStepThreads * i] = sv[i];
    };
    fetch(0);
    commit(0);
    __syncthreads();
    // LDS byte address of the top-left corner of this thread's 25 x 25 tap window: every tap offset is a non-negative immediate
    const unsigned win0 = (unsigned)(size_t)(lds_void_p *)&tile[0][0] + (unsigned)((ty * kLW + tx) * 8);
    const int npairs = (live + 1) >> 1;
    for (int q = 0; q < npairs; q++) {
        const unsigned p0 = plane_off(2 * q), p1 = plane_off(2 * q + 1);
        f32x2 far[8];
#pragma unroll
        for (int t = 0; t < 8; t++) far[t] = (f32x2){ld(rs_src, ofar[t], p0), ld(rs_src, ofar[t], p1)};         // the gathers go first: their latency hides under the LDS taps
        const bool more = q + 1 < npairs;                                              // (wave-uniform)
        if (more) fetch(q + 1);                                                        // next pair's tile: in flight during this pair's taps
        const unsigned win = win0 + (unsigned)((q & 1) * (NE * kStepThreads * 8));
        f32x2 acc = {0.0f, 0.0f};
        // taps in groups of four (half a dilation), two groups in flight: group G = 2 di + hf covers t = 4 hf .. 4 hf + 3 of dilation di
        f32x2 m[2][4];
#define COSA_PAR_TAPS(G)                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                                    \
            const int di = (G) >> 1, t = 4 * ((G) & 1) + u;                                                                \
            const int tt = t < 4 ? t : t + 1;                                                                              \
            const int dy = tt / 3 - 1, dx = tt % 3 - 1;                                                                    \
            const int off = ((kHalo + dy * DIL::d[di]) * kLW + kHalo + dx * DIL::d[di]) * 8;                               \
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(m[(G) & 1][u]) : "v"(win), "i"(off));                       \
        }
        // asm results are not tracked by the compiler's wait insertion: the wait names the registers it covers
#define COSA_PAR_WAIT(cnt, g) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(m[g][0]), "+v"(m[g][1]), "+v"(m[g][2]), "+v"(m[g][3]))
        constexpr int NG = 2 * (ND - 1);
        COSA_PAR_TAPS(0);
#pragma unroll
        for (int G = 0; G < NG; G++) {
            if (G + 1 < NG) {
                COSA_PAR_TAPS(G + 1);
                COSA_PAR_WAIT(4, G & 1);          // this group's taps (issued before the next group's 4 reads) have arrived; the newer ones stay in flight
            } else {
                COSA_PAR_WAIT(0, G & 1);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) mul_add(acc, m[G & 1][u], 4 * G + u);
        }
#undef COSA_PAR_WAIT
#undef COSA_PAR_TAPS
#pragma unroll
        for (int t = 0; t < 8; t++) mul_add(acc, far[t], (ND - 1) * 8 + t);
        if (valid) {
            const float r0 = acc[0], r1 = acc[1];        // (bit_cast of a vector ELEMENT lvalue reads element 0 whatever the index: copy out first)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r0), rs_dst, pix * 4, p0, 0);
            if (2 * q + 1 < live) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r1), rs_dst, pix * 4, p1, 0);
        }
        if (more) commit((q + 1) & 1);                                         // the other buffer: nobody reads it during this iteration
        __syncthreads();                                                       // next tile staged, this one no longer read
    }
}


}  // namespace

int par_make_plan(const int *dilations, int n_dil, ParPlan *plan)
{
    COSA_REQUIRE(dilations && n_dil > 0 && n_dil <= kMaxDil, "PAR: 1..%d dilations supported", kMaxDil);
    plan->n_dil = n_dil;
    const int NN = n_dil * 8;
    float pos[kMaxDil * 8];
    const float sq2 = (float)std::sqrt(2.0);
    for (int di = 0; di < n_dil; di++) {
        COSA_REQUIRE(dilations[di] > 0, "PAR: dilation must be positive");
        plan->dil[di] = dilations[di];
        for (int t = 0; t < 8; t++) {
            const float kk = (t == 0 || t == 2 || t == 5 || t == 7) ? sq2 : 1.0f;
            pos[di * 8 + t] = kk * (float)dilations[di];
        }
    }
    for (int di = n_dil; di < kMaxDil; di++) plan->dil[di] = 1;
    // spec P (position prior): every op one binary32 operation, same order as the oracle
    volatile float sum = 0.0f;
    for (int n = 0; n < NN; n++) sum = sum + pos[n];
    volatile float mean = sum / (float)NN;
    volatile float var = 0.0f;
    for (int n = 0; n < NN; n++) {
        volatile float dl = pos[n] - mean;
        volatile float sq = dl * dl;
        var = var + sq;
    }
    var = var / (float)(NN - 1);
    volatile float sd = std::sqrt((float)var);
    float mx = -INFINITY;
    for (int n = 0; n < NN; n++) {
        volatile float q = pos[n] / (sd + 1e-8f);
        q = q / 0.3f;
        volatile float qq = q * q;
        pos[n] = -qq;
        if (pos[n] > mx) mx = pos[n];
    }
    volatile float es = 0.0f;
    for (int n = 0; n < NN; n++) {
        plan->posw[n] = host_expf(pos[n] - mx);
        es = es + plan->posw[n];
    }
    for (int n = 0; n < NN; n++) plan->posw[n] = plan->posw[n] / es;
    for (int n = NN; n < kMaxDil * 8; n++) plan->posw[n] = 0.0f;
    return COSA_OK;
}

int par_launch_affinity(const float *imgs, float *aff, int B, int h, int w, const ParPlan &plan, hipStream_t st)
{
    bool named = plan.n_dil == Dil6a::n;
    for (int i = 0; named && i < Dil6a::n; i++) named = plan.dil[i] == Dil6a::d[i];
    if (named) {                          // the configuration the reference names: LDS-tiled kernel
        const int tiles_x = (w + kATile - 1) / kATile, tiles_y = (h + kATile - 1) / kATile;
        hipLaunchKernelGGL(par_affinity_tiled_kernel<Dil6a>, dim3(tiles_x * tiles_y, B), dim3(256), 0, st, imgs, aff, h, w, plan, tiles_x);
    } else                                // any other dilation list: the generic kernel (same bits, slower)
        hipLaunchKernelGGL(par_affinity_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, st, imgs, aff, h, w, plan);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// planes per image = Kmax; when kcount != null the live planes of image b are the first kcount[b] planes of each of
// `halves` equal parts of the stack (cam2mask: hi stack then lo stack of every CAM set); halves = 1 when kcount is null.
int par_launch_step(const float *aff, const float *src, float *dst, int B, int Kmax, const int *kcount, int halves,
                    size_t plane_stride, int h, int w, const ParPlan &plan, hipStream_t st)
{
    const int half_planes = Kmax / halves;
    COSA_REQUIRE((size_t)Kmax * h * w < (1ull << 31), "PAR: the planes of one image must stay below 2^31 elements");
    bool named = plan.n_dil == Dil6::n;
    for (int i = 0; named && i < Dil6::n; i++) named = plan.dil[i] == Dil6::d[i];
    const int pin = (B % 8 == 0 || B >= 24) ? 1 : 0;                  // balanced (or nearly) image count per XCD
    const long long images = pin ? 8ll * ((B + 7) / 8) : (long long)B;
    // (the tiled kernel addresses its operands through 32-bit buffer offsets)
    named = named && (size_t)B * plane_stride * sizeof(float) < (1ull << 32) && (size_t)B * Dil6::n * 8 * h * w * sizeof(float) < (1ull << 32);
    if (named) {                          // LDS-tiled step for the named configuration
        const int tiles_x = (w + kTW - 1) / kTW, tiles_y = (h + kTH - 1) / kTH;
        const long long nblk = images * tiles_x * tiles_y;
        COSA_REQUIRE(nblk < 0x7fffffffll, "PAR: grid too large");
        hipLaunchKernelGGL(par_step_tiled_kernel<Dil6>, dim3((unsigned)nblk), dim3(kStepThreads), 0, st, aff, src, dst, kcount, Kmax, halves,
                           half_planes, plane_stride, h, w, B, tiles_x, tiles_y, pin);
    } else {                              // any other dilation list: one pixel per thread, gathers
        const int pix_blocks = (h * w + 255) / 256, groups = (Kmax + 15) / 16;
        const long long nblk = images * pix_blocks * groups;
        COSA_REQUIRE(nblk < 0x7fffffffll, "PAR: grid too large");
        hipLaunchKernelGGL(par_step_kernel, dim3((unsigned)nblk), dim3(256), 0, st, aff, src, dst, kcount, Kmax, halves, half_planes,
                           plane_stride, h, w, plan, B, pix_blocks, groups, pin);
    }
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

}  // namespace cosa

using namespace cosa;

extern "C" size_t cosa_par_workspace_bytes(int B, int K, int h, int w, int n_dil)
{
    const size_t hw = (size_t)h * w;
    return align_up((size_t)B * n_dil * 8 * hw * sizeof(float), 256) + align_up((size_t)B * K * hw * sizeof(float), 256);
}

extern "C" int cosa_par_forward(const float *imgs, const float *masks, float *out, int B, int K, int h, int w,
                                const int *dilations, int n_dil, int num_iter,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(imgs && masks && out && workspace, "cosa_par_forward: null pointer");
    COSA_REQUIRE(B > 0 && K > 0 && h > 0 && w > 0 && B <= 65535, "cosa_par_forward: bad shape");
    COSA_REQUIRE(num_iter >= 0, "cosa_par_forward: num_iter < 0");
    COSA_REQUIRE(out != masks, "cosa_par_forward: out must not alias masks");
    ParPlan plan;
    int rc = par_make_plan(dilations, n_dil, &plan);
    if (rc) return rc;
    if (workspace_bytes < cosa_par_workspace_bytes(B, K, h, w, n_dil)) {
        set_error("cosa_par_forward: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    const size_t hw = (size_t)h * w;
    Carver cv(workspace);
    float *aff = cv.take<float>((size_t)B * n_dil * 8 * hw);
    float *tmp = cv.take<float>((size_t)B * K * hw);
    if (num_iter == 0) {
        if (out != masks) COSA_HIP_CHECK(hipMemcpyAsync(out, masks, (size_t)B * K * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
        return COSA_OK;
    }
    rc = par_launch_affinity(imgs, aff, B, h, w, plan, st);
    if (rc) return rc;
    // ping-pong so that the last step lands in `out`: steps alternate tmp/out
    const float *src = masks;
    for (int it = 0; it < num_iter; it++) {
        float *dst = ((num_iter - 1 - it) & 1) ? tmp : out;
        rc = par_launch_step(aff, src, dst, B, K, nullptr, 1, (size_t)K * hw, h, w, plan, st);
        if (rc) return rc;
        src = dst;
    }
    return COSA_OK;
}
