// gemm_kernels.hip -- op16 MFMA GEMM with fused epilogues for the ViT projections (gfx950).
//
//   Y[M,N] = X[M,K] * W[N,K]^T + bias[N]      (nn.Linear layout: both operands K-contiguous)
//   epilogues:  EPI_BIAS      -> op16 Y                       (qkv projection,          vit.py:121)
//               EPI_GELU      -> op16 gelu_erf(Y)             (mlp.fc1 + GELU,          vit.py:97-98)
//               EPI_RESIDUAL  -> fp32 Yres = R + Y            (attn.proj / mlp.fc2 + residual, vit.py:156-157)
//
// CDNA4 structure: 128x128 output tile per 256-thread workgroup (4 wave64 as 2x2, each 64x64 =
// 4x4 v_mfma_f32_16x16x32_bf16 accumulators), BK = 64, two LDS stages filled by
// global_load_lds_dwordx4 (no VGPR round trip): the LDS image is lane-linear, so the XOR swizzle
// (16-B slot ^ (row & 7), conflict-free ds_read_b128) is applied to the per-lane SOURCE address.
// W rows feed the MFMA A operand and X rows the B operand, so each lane ends up with four
// consecutive output features of one token; the tile is staged through LDS once and leaves as
// whole 256-B rows.  Workgroup ids are remapped so that the tiles sharing an X panel run on the
// same XCD (private L2).
#include "kernels.hpp"
#include "op16.hpp"
#include "c8.hpp"
#include "c4.hpp"
#include <cstdlib>

namespace cosa {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;          // one operand tile: 128 rows x 128 B
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // W tile + X tile
constexpr int CT_LD = 272;                        // bytes per row of the epilogue tile (256 + 16 pad)

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESIDUAL = 2 };

// erf-GELU (nn.GELU default).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below op16 resolution) on the
// hardware exp2 / rcp units: ~14 VALU ops instead of libm erff's ~40 with branches.
__device__ __forceinline__ float gelu_erf(float x)
{
    const float z = __builtin_fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = __builtin_fmaf(-p, e, 1.0f);
    const float erfx = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erfx);
}

// A value that is about to be split into hi = op16(v) and lo = op16(v - hi) must be ONE rounded fp32 number.  Under -ffp-contract=fast the fp16
// build may fuse the multiply that produced v into a conversion (v_fma_mixlo_f16: one rounding of the exact product) at one use and not at
// another (v_mul_f32 + v_cvt: two roundings); on a double-rounding tie the stored hi and the hi the lo half was formed from then differ by
// one fp16 ulp (found in round 6 in the fp16x3 attention: 2 to 4 elements in 150 000).  The empty asm emits no instruction; it makes the
// values opaque registers.  (Its operands are VALU results -- GELU of an accumulator -- never MFMA accumulators themselves.)
#define COSA_SPLIT_OPAQUE(a, b) asm volatile("" : "+v"(a), "+v"(b))

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// one operand tile (128 rows x 64 k) -> LDS, 16 B per lane, 4 wave-instructions per wave
__device__ __forceinline__ void stage_tile(const op16 *__restrict__ src, int row0, int nrows, int ld, int k0,
                                           unsigned char *lds_tile, int wave, int lane)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q = wave * 4 + i;                  // 1-KiB piece: rows 8q .. 8q+7
        const int row = 8 * q + (lane >> 3);
        const int s = (lane & 7) ^ (row & 7);        // logical 16-B slot that lands in physical slot lane&7
        int grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;      // clamp the M tail (masked at the store)
        const op16 *g = src + (size_t)grow * ld + k0 + s * 8;
        __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)(lds_tile + q * 1024), 16, 0, 0);
    }
}

// ---- bf16x3 ("split") operands --------------------------------------------------------------------------------------------------
// A value v is carried as TWO bf16 halves hi = bf16(v), lo = bf16(v - hi) (16 significant bits together), and a product is the three
// MFMA terms hi*hi + lo*hi + hi*lo accumulated in fp32 (the lo*lo term is below 2^-17 relative).  The GEMM kernels do this WITHOUT a
// second code path in their main loops: operand rows are stored as [hi (K) | lo (K) | aug (64)] (row stride 2K + 64), and the K loop
// simply runs over 3K/64 + 1 tiles whose X / W column blocks are picked by split_tile() below:
//     tiles [0, Kp)        X hi  x W hi          tiles [Kp, 2Kp)     X lo x W hi
//     tile   2Kp           X aug x W aug         tiles (2Kp, 3Kp]    X hi x W lo
// The augmentation block carries the BIAS at 16-bit precision with no extra instruction: X aug = (1, 1, 0, ...), W aug[n] = (bias_hi[n],
// bias_lo[n], 0, ...), so the kernel's own bias operand is zero.  Outputs that feed another split GEMM / the split attention kernel are
// written as [hi (N) | lo (N)] with row stride ldy (the consumer's 2N + 64 or 2N); fp32 (residual) outputs are unchanged.
struct SplitGeom { int Kp; };       // K tiles per half (K / 64)
__device__ __forceinline__ int split_tile_x(int kt, int Kp) { return kt <= 2 * Kp ? kt : kt - 2 * Kp - 1; }
__device__ __forceinline__ int split_tile_w(int kt, int Kp) { return kt < Kp ? kt : (kt < 2 * Kp ? kt - Kp : (kt == 2 * Kp ? kt : kt - Kp - 1)); }

// ---- fp16c8 operands (c8.hpp): fp16 hi x fp16 hi on the 16-bit MFMA + two 8-bit correction terms on the block-scaled MFMA ----------------
// Row layout in 128-byte column tiles (Kp = K / 64, Kh = Kp / 2): hi [0, Kp) | lo8 [Kp, Kp + Kh) | hi8 [Kp + Kh, 2 Kp) | aug 2 Kp.  K loop:
//     tiles [0, Kp)              X hi  x W hi    (fp16 MFMA, 64 k per tile)
//     tile   Kp                  X aug x W aug   (fp16: the bias at 16-bit precision, as in the bf16x3 layout)
//     tiles (Kp, Kp + Kh]        X lo8 x W hi8   (e5m2 MFMA, 128 k per tile, scale 2^-11)
//     tiles (Kp + Kh, 2 Kp]      X hi8 x W lo8   (e5m2, scale 2^-11)
// 2 Kp + 1 tiles of equal MFMA time (a 16x16x128 e5m2 MFMA takes the cycles of two 16x16x32 fp16 ones): 2.08x the 1x path, bf16x3 is 3.08x.
// Both 8-bit operands of a tile are fetched with the SAME lane -> byte map (two 16-byte LDS reads per row: slots fq and fq + 4), so byte p of
// lane group g meets byte p of lane group g whatever k index the hardware gives it, and the E8M0 scales are uniform: no dependence on
// the instruction's k layout.
__device__ __forceinline__ bool c8_is_f8(int kt, int Kp) { return kt > Kp; }
__device__ __forceinline__ int c8_tile_x(int kt, int Kp) { return kt < Kp ? kt : (kt == Kp ? 2 * Kp : kt - 1); }                 // hi | aug | lo8 | hi8
__device__ __forceinline__ int c8_tile_w(int kt, int Kp)
{
    const int Kh = Kp >> 1;
    return kt < Kp ? kt : (kt == Kp ? 2 * Kp : (kt <= Kp + Kh ? kt + Kh - 1 : kt - Kh - 1));                                        // hi | aug | hi8 | lo8
}
typedef int c8_i32x8 __attribute__((ext_vector_type(8)));
typedef int c8_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c8_i32x8 c8_cat(op16x8 lo, op16x8 hi)
{
    return __builtin_shufflevector(__builtin_bit_cast(c8_i32x4, lo), __builtin_bit_cast(c8_i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
constexpr int kC8ScaleW = 0x7f7f7f7f;       // E8M0 2^0 in every byte (A operand: weight rows)
constexpr int kC8ScaleX = 0x74747474;       // E8M0 2^-11 (B operand: token rows): both correction terms carry one lo8 factor = x 2^11
#define COSA_MFMA_C8(a8, b8, c) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 1, 1, 0, kC8ScaleW, 0, kC8ScaleX)

// ---- fp16c4 operands (c4.hpp): fp16 hi x fp16 hi on the 16-bit MFMA + BOTH correction terms as one stream of FP4 (e2m1) MX blocks ------
// Row layout in 128-byte column tiles (Kp = K / 64, Kq = K / 128): hi [0, Kp) | c4 [Kp, Kp + Kq) | (unused) | aug 2 Kp.  K loop:
//     tiles [0, Kp)              X hi  x W hi    (fp16 MFMA, 64 k per tile)
//     tile   Kp                  X aug x W aug   (fp16: the bias)
//     tiles (Kp, Kp + Kq]        X c4  x W c4    (e2m1 MFMA: a 128-byte row of a tile = 8 blocks of [16 lo' | 16 hi] (X) / [16 hi | 16 lo'] (W);
//                                                 a lane's 16-byte fragment of k-step ks is block fq + 4 ks = the 32 values of ONE instruction)
// Kp + 1 + Kq tiles of equal MFMA time (a 16x16x128 e2m1 MFMA takes the cycles of one 16x16x32 fp16 one): 1.58x the plain path at K = 768,
// fp16c8 is 2.08x.  The E8M0 scale of every (row, block) comes from the operand's scale tensor through a small LDS ring (see the kernel).
template <int OA, int OB>
__device__ __forceinline__ f32x4 mfma_c4(c8_i32x8 a4, c8_i32x8 b4, f32x4 c, int sa, int sb)
{
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, c, 4, 4, OA, sa, OB, sb);
}
__device__ __forceinline__ c8_i32x8 c4_op(op16x8 v)
{
    const c8_i32x4 t = __builtin_bit_cast(c8_i32x4, v);
    return (c8_i32x8){t[0], t[1], t[2], t[3], 0, 0, 0, 0};           // (the backend keeps only the four registers the fp4 format reads)
}
// maximum over the four lanes lane, lane ^ 16, lane ^ 32, lane ^ 48 of a non-negative float (VALU lane swaps, no LDS traffic)
__device__ __forceinline__ float c4_max4(float v)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned m = a[0] > a[1] ? a[0] : a[1];                     // (non-negative floats order like their bit patterns)
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return __builtin_bit_cast(float, b[0] > b[1] ? b[0] : b[1]);
}
constexpr int kC4ScaleLds = 16384;          // two 8-KB chunks (X and W scales of two c4 tiles each) behind the 128-KB operand ring
// Round 5: the persistent kernel's output leaves through per-wave LDS staging rows behind the ring (and the scale ring): [16 tokens][128 B],
// 2 KB per plane and wave, two planes unless fp16c4 operands (whose scale ring takes the other 16 KB): 160 KB of LDS in either case.
constexpr int kStageLdsC4 = 16384, kStageLds = 32768;

template <int EPI, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const op16 *__restrict__ X, const op16 *__restrict__ W,
                                                          const op16 *__restrict__ bias, const float *__restrict__ R,
                                                          void *__restrict__ Yv, int M, int N, int K, int tiles_m, int tiles_n,
                                                          int ld, int ldy, void *__restrict__ Y2v, int tail_o0 = -1)
{
    // SPLIT: K is the LOGICAL contraction length; operand rows have stride ld = 2K + 64, 16-bit outputs go to [hi | lo] rows of stride ldy
    // SPLIT == 3 (fp16c8 operands): tail mode with the residual epilogue only -- the leftover jobs of the persistent kernel's output projection
    // (N = 768: 1032 jobs = 4 x 256 + 8), same tile order and MFMA sequence per output element, hence the same bits.  (Measured and not kept for
    // fp16c4's fc2, K = 3072: a 128 x 128 quarter-job pays ~1.35 us per tile -- 8 LDS-DMA issues per wave and tile, whatever the ring depth --,
    // so its 73 tiles take 99 us, as long as the lone fifth round of the persistent kernel they would replace.)
    static_assert(SPLIT < 3 || (SPLIT == 3 && EPI == EPI_RESIDUAL), "fp16c8 operands: residual epilogue (tail mode) only; fp16c4: not supported here");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int m0, n0;
    if (tail_o0 >= 0) {
        // tail of the persistent kernel's job list (cosa_gemm_bf16): job o = tail_o0 + blockIdx.x / 4 of ITS enumeration (tiles_m x tiles_n
        // are its 256 x 256 tile counts here), one 128 x 128 quarter per workgroup
        const int ntl = tiles_m * tiles_n, cq = ntl >> 3, cr = ntl & 7;
        const int o = tail_o0 + ((int)blockIdx.x >> 2), sub = blockIdx.x & 3;
        const int xcd = o & 7, idx = o >> 3;
        const int t = (xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq) + idx;
        const int tm = t / tiles_n;
        m0 = tm * 256 + (sub >> 1) * BM;
        n0 = (t - tm * tiles_n) * 256 + (sub & 1) * BN;
        if (m0 >= M) return;                                            // lower half of a partial last m-panel
    } else {
        // XCD-aware remap (bijective): consecutive ids of one XCD walk the n-tiles of one m-panel
        const int nwg = tiles_m * tiles_n;
        int wg = blockIdx.x;
        {
            const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
            wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
        m0 = tm * BM;
        n0 = tn * BN;
    }
    const int wr = wave >> 1, wc = wave & 1;        // wave tile: n-rows [wr*64,+64) x m-cols [wc*64,+64)

    f32x4 acc[4][4];
    const bool tail = tail_o0 >= 0;
    // tail mode reproduces the persistent kernel's summation order bit for bit (its jobs and these must be interchangeable: a token's
    // result may not depend on which of the two computed it): the accumulation STARTS from the bias (bf16 out) or from the fp32 residual
    // tile (residual epilogue, bias added last), then the products in ascending k
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tail) {
                const int nl = (wave >> 1) * 64 + 16 * i + 4 * (lane >> 4), ml = (wave & 1) * 64 + 16 * j + (lane & 15);
                if (EPI == EPI_RESIDUAL) {
                    if (m0 + ml < M) acc[i][j] = *reinterpret_cast<const f32x4 *>(R + (size_t)(m0 + ml) * N + n0 + nl);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[i][j][r] = (float)bias[n0 + nl + r];
                }
            }
        }

    const int Kp = K / BK;
    const int nk = SPLIT == 1 ? 3 * Kp + 1 : (SPLIT == 3 ? 2 * Kp + 1 : Kp);
    auto tile_x = [&](int kt) { return SPLIT == 1 ? split_tile_x(kt, Kp) : (SPLIT == 3 ? c8_tile_x(kt, Kp) : kt); };
    auto tile_w = [&](int kt) { return SPLIT == 1 ? split_tile_w(kt, Kp) : (SPLIT == 3 ? c8_tile_w(kt, Kp) : kt); };
    stage_tile(W, n0, N, ld, 0, smem, wave, lane);
    stage_tile(X, m0, M, ld, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        unsigned char *cur = smem + (kt & 1) * STAGE_BYTES;
        unsigned char *nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        if (kt + 1 < nk) {
            stage_tile(W, n0, N, ld, tile_w(kt + 1) * BK, nxt, wave, lane);
            stage_tile(X, m0, M, ld, tile_x(kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const unsigned char *At = cur + (wr * 64) * 128;
        const unsigned char *Bt = cur + TILE_BYTES + (wc * 64) * 128;
        if (SPLIT == 3 && kt > Kp) {
            // e5m2 correction tiles: one 16x16x128 MFMA per accumulator on both 16-byte fragments of a row (as the persistent kernel)
            op16x8 a2[4][2], b2[4][2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = i * 16 + frow;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int slot = ((fq + 4 * ks) ^ (row & 7)) << 4;
                    a2[i][ks] = *reinterpret_cast<const op16x8 *>(At + row * 128 + slot);
                    b2[i][ks] = *reinterpret_cast<const op16x8 *>(Bt + row * 128 + slot);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c8_i32x8 a8 = c8_cat(a2[i][0], a2[i][1]);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = COSA_MFMA_C8(a8, c8_cat(b2[j][0], b2[j][1]), acc[i][j]);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            op16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = i * 16 + frow;
                const int slot = ((fq + 4 * ks) ^ (row & 7)) << 4;
                a[i] = *reinterpret_cast<const op16x8 *>(At + row * 128 + slot);
                b[i] = *reinterpret_cast<const op16x8 *>(Bt + row * 128 + slot);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = COSA_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: bias (+GELU) in registers -> LDS tile [m][n] -> whole-row stores (+fp32 residual) ----
    // acc[i][j][r]: n = wr*64 + 16i + 4*fq + r,  m = wc*64 + 16j + frow
    unsigned char *Ct = smem;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int nl = wr * 64 + 16 * i + 4 * fq;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) bv[r] = (tail && EPI != EPI_RESIDUAL) ? 0.0f : (float)bias[n0 + nl + r];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ml = wc * 64 + 16 * j + frow;
            if (EPI == EPI_RESIDUAL) {
                // keep fp32: two 128x64 fp32 halves would not fit the 2-stage LDS at once -> write bf16x... use f32 tile rows of 128 n
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = acc[i][j][r] + bv[r];
                *reinterpret_cast<f32x4 *>(Ct + ml * (BN * 4 + 16) + nl * 4) = v;
            } else {
                op16x4 v, vlo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float t = acc[i][j][r] + bv[r];
                    if (SPLIT == 2) {                         // dual: pre-activation | gelu
                        v[r] = (op16)t;
                        vlo[r] = (op16)gelu_erf(t);
                    } else {
                        if (EPI == EPI_GELU) t = gelu_erf(t);
                        if (SPLIT) asm volatile("" : "+v"(t));      // a ROUNDED fp32 value before it is split (COSA_SPLIT_OPAQUE below)
                        v[r] = (op16)t;
                        if (SPLIT) vlo[r] = (op16)(t - (float)v[r]);
                    }
                }
                *reinterpret_cast<op16x4 *>(Ct + ml * CT_LD + nl * 2) = v;
                if (SPLIT) *reinterpret_cast<op16x4 *>(Ct + BM * CT_LD + ml * CT_LD + nl * 2) = vlo;
            }
        }
    }
    __syncthreads();
    if (EPI == EPI_RESIDUAL) {
        float *Y = static_cast<float *>(Yv);
        // 128 rows x 512 B: 4096 16-B chunks, 16 per thread
#pragma unroll
        for (int c = tid; c < BM * 32; c += 256) {
            const int ml = c >> 5, s = c & 31;
            if (m0 + ml < M) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + ml * (BN * 4 + 16) + s * 16);
                const size_t o = (size_t)(m0 + ml) * N + n0 + s * 4;
                if (tail) {
                    *reinterpret_cast<f32x4 *>(Y + o) = v;                       // the residual is already in the accumulators
                } else {
                    const f32x4 rv = *reinterpret_cast<const f32x4 *>(R + o);
                    *reinterpret_cast<f32x4 *>(Y + o) = v + rv;
                }
            }
        }
    } else {
        op16 *Y = static_cast<op16 *>(Yv);
#pragma unroll
        for (int c = tid; c < BM * 16; c += 256) {
            const int ml = c >> 4, s = c & 15;
            if (m0 + ml < M) {
                *reinterpret_cast<uint4 *>(Y + (size_t)(m0 + ml) * ldy + n0 + s * 8) =
                    *reinterpret_cast<const uint4 *>(Ct + ml * CT_LD + s * 16);
                if (SPLIT == 1)
                    *reinterpret_cast<uint4 *>(Y + (size_t)(m0 + ml) * ldy + N + n0 + s * 8) =
                        *reinterpret_cast<const uint4 *>(Ct + BM * CT_LD + ml * CT_LD + s * 16);
                if (SPLIT == 2)
                    *reinterpret_cast<uint4 *>(static_cast<op16 *>(Y2v) + (size_t)(m0 + ml) * ldy + n0 + s * 8) =
                        *reinterpret_cast<const uint4 *>(Ct + BM * CT_LD + ml * CT_LD + s * 16);
            }
        }
    }
}


// =====================================================================================================
// The phase stream of the 256 x 256 x 64 kernel: 8 waves, FOUR-PHASE-PER-K-TILE ping-pong schedule (MI355X guide, "256^2 8-phase").
// (Rounds 1-2 kept its measured predecessors -- 256x128 3-stage, 256x256 plain / persistent with an LDS epilogue, and this schedule as a
// one-tile-per-workgroup kernel -- in the library as variants 2-5; round 3 removed them: the persistent kernel below is the one that runs.)
//
// Each operand K-tile is two 16-KB half-tiles (128 rows x 64 k): X0 | W0 | X1 | W1, two K-tiles deep = 128 KB of LDS.
// Wave (wr, wc) = (wave >> 2, wave & 3) owns features  n0 + {0,128} + wr*64 + [0,64)  and tokens
// m0 + {0,128} + wc*32 + [0,32): one 64-row quarter of EACH W half and one 32-row quarter of EACH X half, so every
// half-tile is read in exactly one phase of the K-tile by all waves and can be refilled right after it:
//     phase 1: read X0 (4 ds_read_b128) + W0 (8)   | MFMA W0 x X0 | refill  W1 of tile t+1
//     phase 2: read X1 (4)                         | MFMA W0 x X1 | refill  X0 of tile t+2
//     phase 3: read W1 (8)                         | MFMA W1 x X1 | refill  W0 of tile t+2
//     phase 4: (X0 fragments still in registers)   | MFMA W1 x X0 | refill  X1 of tile t+2, s_waitcnt vmcnt(6)
// Every phase is  [reads + 2 LDS-DMA pieces]  barrier  [16 MFMAs]  barrier, and the wr = 1 waves run one barrier behind
// the wr = 0 waves: the two waves that share a SIMD alternate between the read section and the MFMA section, so the
// matrix pipe always has one of them.  The DMA stream runs 7 half-tiles ahead of the reads; the only vmcnt wait is the
// counted one in phase 4 (3 half-tiles = 6 loads stay in flight), and a buffer is read no earlier than one phase
// after the barrier that follows that wait (two barriers: the staggered group waits one barrier later).
// Rows past M come back as zeros from the buffer range check; K-tiles past K are "loaded" from an out-of-range
// offset (no memory traffic) so that the vmcnt arithmetic is the same in the last tiles.
// =====================================================================================================
constexpr int V5_HALF = 16384;
constexpr int V5_BUF = 4 * V5_HALF;                    // X0 | W0 | X1 | W1
constexpr size_t kLdsBytesV5 = 2 * V5_BUF;             // 128 KB; also holds the 256 x 272-byte epilogue half-tile

#define V5_FENCE() __builtin_amdgcn_sched_barrier(0)
#define V5_BARRIER()                          \
    do {                                      \
        V5_FENCE();                           \
        __builtin_amdgcn_s_barrier();         \
        V5_FENCE();                           \
    } while (0)

// =====================================================================================================
// v6: that phase stream as a PERSISTENT kernel whose epilogue rides inside the next phases.
//
// One workgroup per CU walks its tiles (jobs).  The LDS-DMA stream never stops at a job boundary: the half-tiles
// "7 ahead" simply belong to the next job (its operand panels are described by a second pair of buffer resources),
// so no job after the first pays the pipeline fill.  A 64 x 32 accumulator quadrant is final after the phase that
// last used it, and is written out (bias is already in: the first MFMA of a job takes C = bias; optional GELU;
// v_cvt_pk_bf16; v_permlane16_swap so that a lane owns 8 consecutive features; 16-byte buffer stores straight from
// registers, no LDS) inside the MFMA section of the FOLLOWING phase, in the shadow of MFMAs that work on another quadrant:
//     last K-tile:  phase 2 stores Q(W0,X0), phase 3 Q(W0,X1), phase 4 Q(W1,X1);  first phase of the next job: Q(W1,X0).
// Token rows past M never reach memory: the Y window is a buffer resource whose num_records ends with the last valid
// row, and every row offset is in the per-lane VGPR offset (the part of the address that is range-checked).
// vmcnt arithmetic (loads and stores retire in issue order on gfx9-class vmcnt): the phase-4 wait must leave only the
// operations issued after phase 1's DMA in flight: 6 in the steady state, 14 in a job's last K-tile (3 x 2 DMA + 2 x 4 stores).
// =====================================================================================================
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Output stores use the default cache policy: stand-alone, non-temporal stores are faster (qkv projection: plain 304 us, nt 287 us), but in the
// training step the consumer kernel runs next and wants the tile in L2 / the Infinity Cache (plain 293 us vs 347 us with nt + sc0 + sc1).
// (Rounds 2-3 carried the timing ablations behind these numbers -- store policies, epilogue without stores, an L2-resident store window, banded
// tile orders, start staggers -- as template parameters and environment switches of this kernel; round 4 removed them: HISTORY.md has the numbers.)
// SPLIT: bf16x3 operands (see split_tile_x / split_tile_w above): K is the logical contraction length, operand rows have stride ld
// (= 2K + 64), the K loop has 3K/64 + 1 tiles, 16-bit outputs are written as [hi | lo] halves of rows with stride ldy.
template <int EPI, int SPLIT = 0, int FR = 4>
__global__ __launch_bounds__(512, 1) void gemm_bf16_v6_kernel(const op16 *__restrict__ X, const op16 *__restrict__ W,
                                                             const op16 *__restrict__ bias, const float *__restrict__ R,
                                                             void *__restrict__ Yv, int M, int N, int K, int tiles_m, int tiles_n,
                                                             unsigned long long *__restrict__ stamps, int ld, int ldy, void *__restrict__ Y2v, int ntiles_run,
                                                             const unsigned char *__restrict__ xsc, const unsigned char *__restrict__ wsc, unsigned char *__restrict__ ysc)
{
    // optional device-side span of this launch (100 MHz wall clock; min start / max end over workgroups): HIP events cannot be
    // recorded inside a captured hipGraph on ROCm, so bench.py's roofline leg reads these (cosa_gemm_set_stamp_slot)
    if (stamps && threadIdx.x == 0) atomicMin(&stamps[2 * (blockIdx.x & 63)], __builtin_amdgcn_s_memrealtime());     // 64 shards: one address would serialise the workgroups' atomics
    constexpr bool RES = EPI == EPI_RESIDUAL;          // fp32 out = fp32 residual + X W^T + bias
    constexpr int ES = RES ? 4 : 2;
    constexpr int FL = 0x00020000;
    static_assert(FR == 4 || (FR == 3 && SPLIT == 0), "FR: 16-feature fragments per wave and W half (tile width 64 FR)");
    constexpr bool C8 = SPLIT == 3;                    // fp16c8 operands (see c8_tile_x / c8_tile_w); 16-bit GELU outputs leave as c8 rows
    constexpr bool C8OUT = C8 && EPI == EPI_GELU;
    constexpr bool C4 = SPLIT == 4;                    // fp16c4 operands (c4.hpp); 16-bit GELU outputs leave as c4 rows + scale bytes
    constexpr bool C4OUT = C4 && EPI == EPI_GELU;
    constexpr bool CX = C8 || C4;                      // either: the bias rides in the augmentation tile, two loop bodies
    static_assert(!C4 || FR == 4, "fp16c4: 256-wide jobs only (the scale tensors are laid out for them)");
    constexpr int TN = 64 * FR, HN = 32 * FR, WN = 16 * FR;     // features per tile / per W half / per wave inside a half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    const int Kp = K / BK;
    const int Kq = Kp >> 1;                            // fp16c4: c4 tiles per row
    const int nk = SPLIT == 1 ? 3 * Kp + 1 : (C8 ? 2 * Kp + 1 : (C4 ? Kp + 1 + Kq : Kp));
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    unsigned char *Yb = static_cast<unsigned char *>(Yv);

    auto tile_of = [&](int o, int &m0_, int &n0_) {      // every XCD (o & 7) walks a contiguous chunk of the m-panel-major tile list
        const int xcd = o & 7, idx = o >> 3;
        const int t = (xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq) + idx;
        const int tm = t / tiles_n;
        m0_ = tm * 256;
        n0_ = (t - tm * tiles_n) * TN;
    };
    auto descX = [&](int m0_) {
        int rows = M - m0_;
        rows = rows > 256 ? 256 : rows;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)m0_ * ld), 0, rows * ld * 2, FL);
    };
    auto descW = [&](int n0_) { return __builtin_amdgcn_make_buffer_rsrc((void *)(W + (size_t)n0_ * ld), 0, TN * ld * 2, FL); };
    // SPLIT == 2 ("dual", EPI_GELU only): the pre-activation goes to Y and gelu(.) to Y2 (same [M, N] geometry): the training forward of
    // mlp.fc1 keeps the pre-activation for GELU' without a second pass over it (models/vit/vit.py:96-102 and its autograd)
    auto descY2 = [&](int m0_, int n0_) {
        int rows = M - m0_;
        rows = rows > 256 ? 256 : rows;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(static_cast<unsigned char *>(Y2v) + ((size_t)m0_ * ldy + n0_) * 2), 0, (rows * ldy - n0_) * 2, FL);
    };
    auto descY = [&](int m0_, int n0_) {
        int rows = M - m0_;
        rows = rows > 256 ? 256 : rows;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(Yb + ((size_t)m0_ * ldy + n0_) * ES), 0, (rows * ldy - n0_) * ES, FL);
    };
    // the residual window of a job as raw descriptor words (inline-asm loads), same geometry as the Y window
    auto descR = [&](int m0_, int n0_) {
        int rows = M - m0_;
        rows = rows > 256 ? 256 : rows;
        const unsigned long long p = (unsigned long long)(R + (size_t)m0_ * N + n0_);
        return (u32x4){(unsigned)p, (unsigned)(p >> 32) & 0xffffu, (unsigned)((rows * N - n0_) * 4), (unsigned)FL};
    };
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, 0, FL);   // every access out of range

    // fp16c4: the E8M0 scale bytes of a job's blocks travel through a ring of two 8-KB chunks behind the operand ring; a chunk holds the X and
    // the W scales of two consecutive c4 tiles (2 KB each, in the order the lanes read them: c4.hpp) and is ONE LDS-DMA instruction per wave
    // (waves 0-3: the X half, 4-7: the W half), issued two tiles before its first use, in phase 1 IN FRONT of that phase's operand DMA -- so
    // the counted vmcnt waits, which leave only what was issued after phase 1's operand DMA in flight, retire it without a changed count.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // scalar copy: selects the chunk half's descriptor without a waterfall
    __amdgpu_buffer_rsrc_t sXr = dead, sWr = dead, sYr = dead;
    if (C4) {
        sXr = __builtin_amdgcn_make_buffer_rsrc((void *)xsc, 0, tiles_m * Kq * 2048, FL);
        sWr = __builtin_amdgcn_make_buffer_rsrc((void *)wsc, 0, tiles_n * Kq * 2048, FL);
        if (C4OUT) sYr = __builtin_amdgcn_make_buffer_rsrc((void *)ysc, 0, tiles_m * (N >> 7) * 2048, FL);
    }
    const unsigned sc_rd_x = (unsigned)(((wc * 16 + frow) * 4 + fq) * 8), sc_rd_w = (unsigned)(4096 + ((wr * 16 + frow) * 4 + fq) * 16);
    // c4 rows out (GELU epilogue): after the lane swaps a lane owns block j = wr 4 + (fq & 1) + 2 (fq >> 1) of the tile n0 / 128 + qa
    const unsigned sc_wr_y = (unsigned)((((wc * 16 + frow) * 4 + (fq & 1) + 2 * (fq >> 1)) * 8) + wr);
    u32x2 sxv = {0, 0};
    u32x4 swv = {0, 0, 0, 0};
#define V6_SCDMA(c)                                                                                                    \
    do {                                                                                                               \
        const int c_ = (c);                                                                                            \
        const __amdgpu_buffer_rsrc_t rs_ = wave_u < 4 ? sXr : sWr;                                                     \
        const int so_ = wave_u < 4 ? (((m0 >> 8) * Kq + 2 * c_) * 2048 + wave_u * 1024) : (((n0 >> 8) * Kq + 2 * c_) * 2048 + (wave_u - 4) * 1024); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(smem + kLdsBytesV5 + (c_ & 1) * 8192 + wave_u * 1024), 16, lane * 16, so_, 0, 0); \
    } while (0)
    // the lane's scales of c4 tile q: X: 8 bytes [qb][jj][ks], W: 16 bytes [qa][i][ks]
#define V6_LDSC(q)                                                                                                     \
    do {                                                                                                               \
        const unsigned char *sb_ = smem + kLdsBytesV5 + (((q) >> 1) & 1) * 8192 + ((q) & 1) * 2048;                    \
        sxv = *reinterpret_cast<const u32x2 *>(sb_ + sc_rd_x);                                                         \
        swv = *reinterpret_cast<const u32x4 *>(sb_ + sc_rd_w);          /* (sc_rd_w includes the chunk's 4-KB X half) */  \
    } while (0)

    // job-independent per-lane offsets: DMA source inside a 256-row operand panel (same for X and W when FR == 4) ...
    // FR == 3: a W half has 96 rows; the DMA slots of rows 96 .. 127 of a half get an out-of-range offset (nothing fetched, zeros land
    // in LDS rows that no fragment read touches), so that every wave still issues the same number of VMEM operations
    unsigned vo[2][2], voW[2][2];
    {
        const int sw = ((lane & 7) ^ (lane >> 3)) * 8;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int rl = 8 * (2 * wave + i) + (lane >> 3);
                vo[h][i] = (unsigned)(((h * 128 + rl) * ld + sw) * 2);
                voW[h][i] = FR == 4 ? vo[h][i] : (rl < HN ? (unsigned)(((h * HN + rl) * ld + sw) * 2) : 0x7ffff000u);
            }
    }
    // ... and the store offset inside the tile's Y window for each 16-token row group (b, jj); after the lane swap a lane
    // owns features  wr*64 + pair*32 + (fq&1)*16 + 4*(fq&2) .. +7  of its token
    // (fp32 output: no swap, a lane owns features wr*64 + ii*16 + 4fq .. +3)
    // (one register: the offsets of the other three row groups are formed at their use by a volatile add of a scalar -- V6_VOY below; kept as four
    // loop-invariant registers they were what the GELU / bf16x3 instantiations spilled)
    const unsigned voY0 = RES ? (unsigned)(((wc * 32 + frow) * ldy + wr * WN + 4 * fq) * 4)
                              : (unsigned)(((wc * 32 + frow) * ldy + wr * WN + (fq & 1) * 16 + 4 * (fq & 2)) * 2);
    // FR == 3, 16-bit output: the third fragment of a wave has no partner to swap with; its lane keeps features 4fq .. 4fq + 3 (8-byte stores)
    const unsigned voY3 = (unsigned)((32 + 4 * fq - (fq & 1) * 16 - 4 * (fq & 2)) * 2);
    // c8 rows out: the lane's feature offset inside the tile in BYTES of an 8-bit plane (voY holds it doubled, for the fp16 plane)
    // (the combined 16-byte store: byte offset of the lane's 16 features in an 8-bit plane, minus twice its offset in the fp16 plane, which voY carries)
    const unsigned fo16m = (unsigned)((wr * WN + 16 * (fq & 1) + 32 * (fq >> 1)) - 2 * (wr * WN + (fq & 1) * 16 + 4 * (fq & 2)));

    // ---- coalesced output stores (round 5) ------------------------------------------------------------------------------------------
    // An accumulator fragment has the token on lane & 15 and four features per lane: stored from the registers, ONE buffer_store_b128
    // touches 16 token rows and leaves the CU as 64 separate 16-byte write requests (lanes of one row are 16 apart: nothing coalesces).
    // Measured in round 3 as "the stores cost what their bytes take at HBM speed, with no overlap" (HISTORY section 7): it is the request
    // count -- the store path of a CU took ~70 cycles per such instruction, and the LDS-DMA operand stream waits behind it.  Now a wave
    // transposes 16 tokens x 128 B through 2 KB of its own LDS (one ds_write_b128 / ds_read_b128 pair per former store, 16-byte chunks
    // XOR-ed by the row: conflict-free both ways; no barrier, a wave's LDS operations complete in order) and the same NUMBER of stores
    // (the counted vmcnt waits are unchanged) each write 8 rows x 128 contiguous bytes: 8 requests instead of 64.
    // Measured (tools/ab_gemm_libs.py, interleaved, same box): qkv -4 %, output projection / fc2 (fp32 rows) -2...5 %, fp16c4 qkv / fc2 -2...3 %;
    // the GELU epilogues got 5 % SLOWER (their VALU work is the long pole of those phases and the LDS round trip adds to it): they keep the
    // register stores, and so do the 192-wide jobs (their third fragment has no partner).
    constexpr bool STG = FR == 4 && EPI != EPI_GELU;
    constexpr int STGW = C4 ? 2048 : 4096;              // staging bytes per wave: one plane, or two (dual / bf16x3 outputs)
    const unsigned stg0 = (unsigned)(kLdsBytesV5 + (C4 ? kC4ScaleLds : 0) + wave * STGW);
    // write: token frow, 16-byte chunk c of the 128-byte row -- 16-bit outputs: c = 4 pr + 2 (fq & 1) + (fq >> 1) (8 features per lane after
    // the lane swap), fp32 outputs: c = 4 (ii & 1) + fq (4 features per lane); pr / (ii & 1) toggle byte bit 6
    const unsigned stg_w = stg0 + frow * 128 + ((((RES ? fq : (fq & 1) * 2 + (fq >> 1))) ^ (frow & 7)) << 4);
    // read: lane l takes chunk l & 7 of row (l >> 3) (+ 8 for the second read), i.e. 8 lanes = one 128-byte row segment of the output
    const unsigned stg_r = stg0 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned voS = (unsigned)(((wc * 32 + (lane >> 3)) * ldy + wr * WN) * ES + (lane & 7) * 16);
#define V6_STG_WR(u4, e, plane) *reinterpret_cast<u32x4 *>(smem + ((stg_w ^ ((e) * 64)) + (plane) * 2048)) = (u4)
#define V6_STG_RD(it, plane) (*reinterpret_cast<const u32x4 *>(smem + (stg_r + (it) * 1024 + (plane) * 2048)))
    // store offset of rows (qb, jj, it): the row part belongs in the VGPR offset (the part of the address the range check sees).  The add is
    // volatile asm on a scalar product: left to the compiler, the eight loop-invariant sums are hoisted out of the job loop, spilled, and
    // re-loaded from scratch behind `s_waitcnt vmcnt(0)` -- in the middle of the counted LDS-DMA stream.
    const int ldy_es = __builtin_amdgcn_readfirstlane(ldy * ES);
#define V6_VOY(g4)                                                                                                   \
    ({                                                                                                               \
        unsigned r_ = voY0;                                                                                          \
        if ((g4) != 0)                                                                                               \
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(r_) : "s"((((g4) >> 1) * 128 + ((g4) & 1) * 16) * ldy_es), "v"(voY0));  \
        r_;                                                                                                          \
    })
#define V6_STG_OFF(qb, jj, it)                                                                                       \
    ({                                                                                                               \
        unsigned r_ = voS;                                                                                           \
        if ((qb) * 128 + (jj) * 16 + (it) * 8 != 0)                                                                  \
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(r_) : "s"(((qb) * 128 + (jj) * 16 + (it) * 8) * ldy_es), "v"(voS));  \
        r_;                                                                                                          \
    })

    int o = blockIdx.x;
    int m0, n0, m1 = 0, n1 = 0, pn0 = 0, pm0 = 0;
    tile_of(o, m0, n0);
    bool has_next = o + G < ntiles_run;      // (ntiles_run <= ntiles: the jobs past it are left to a tail launch, cosa_gemm_bf16)
    if (has_next) tile_of(o + G, m1, n1);
    __amdgpu_buffer_rsrc_t cX = descX(m0), cW = descW(n0), cY = descY(m0, n0), pY = cY;
    __amdgpu_buffer_rsrc_t cY2 = SPLIT == 2 ? descY2(m0, n0) : cY, pY2 = cY2;
    __amdgpu_buffer_rsrc_t nX = has_next ? descX(m1) : dead, nW = has_next ? descW(n1) : dead;
    u32x4 cR = {0, 0, 0, 0}, nR = {0, 0, 0, 0};
    if (RES) {
        cR = descR(m0, n0);
        if (has_next) nR = descR(m1, n1);
    }
    bool have_prev = false;

    // bias of the job about to start, as packed op16: features a*128 + wr*64 + ii*16 + 4fq .. +3.  Loaded by hand (inline asm) so
    // that the compiler does not put its own vmcnt wait in front of the first use: the phase-4 waits cover these loads.
    u32x2 bb[2][4];
    const unsigned bias_lane = (unsigned)((wr * WN + 4 * fq) * 2);        // scalar base + 32-bit lane offset: no 64-bit per-lane pointer to keep alive
#define V6_LOAD_BIAS(nbase)                                                                               \
    if (!CX) _Pragma("unroll") for (int a_ = 0; a_ < 2; a_++) _Pragma("unroll") for (int i_ = 0; i_ < FR; i_++) {  \
        const op16 *p_ = bias + (nbase) + a_ * HN + i_ * 16;                                              \
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(bb[a_][i_]) : "v"(bias_lane), "s"(p_) : "memory"); \
    }

    // which: 0 X0, 1 W0, 2 X1, 3 W1; K-tile kt of the running job, or kt - nk of the next one
#define V6_STAGE(which, kt, bsel)                                                                                      \
    do {                                                                                                               \
        const int kt_ = (kt);                                                                                          \
        const bool own_ = kt_ < nk;                                                                                    \
        const __amdgpu_buffer_rsrc_t rs_ = ((which) & 1) ? (own_ ? cW : nW) : (own_ ? cX : nX);                        \
        const int t_ = own_ ? kt_ : kt_ - nk;                                                                          \
        const int so_ = (SPLIT == 1 ? (((which) & 1) ? split_tile_w(t_, Kp) : split_tile_x(t_, Kp))                       \
                         : (C8 ? (((which) & 1) ? c8_tile_w(t_, Kp) : c8_tile_x(t_, Kp)) : (C4 ? c8_tile_x(t_, Kp) : t_))) * 128;   \
        unsigned char *dst_ = smem + (bsel) * V5_BUF + (which) * V5_HALF + (2 * wave) * 1024;                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)dst_, 16, ((which) & 1) ? voW[(which) >> 1][0] : vo[(which) >> 1][0], so_, 0, 0);           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(dst_ + 1024), 16, ((which) & 1) ? voW[(which) >> 1][1] : vo[(which) >> 1][1], so_, 0, 0);  \
    } while (0)

    f32x4 acc[8][4];
    // residual: the accumulators of a quadrant START as the residual tile (so the epilogue has no loads).  The loads are
    // issued by hand right after the quadrant's previous contents were stored, two to three phases before its first MFMA.
#define V6_RLOAD(qa, qb, rsR)                                                                                        \
    _Pragma("unroll") for (int ii = 0; ii < FR; ii++) _Pragma("unroll") for (int jj = 0; jj < 2; jj++)               \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(acc[(qa) * 4 + ii][(qb) * 2 + jj])           \
                     : "v"(V6_VOY((qb) * 2 + jj)), "s"(rsR), "s"(((qa) * HN + ii * 16) * 4) : "memory");

    // pipeline fill (first job only): bias, [residual tile,] K-tile 0 complete, X0 W0 X1 of K-tile 1
    V6_LOAD_BIAS(n0);
    if (RES) {
        V6_RLOAD(0, 0, cR);
        V6_RLOAD(0, 1, cR);
        V6_RLOAD(1, 1, cR);
        V6_RLOAD(1, 0, cR);
    }
    V5_FENCE();
    V6_STAGE(0, 0, 0);
    V6_STAGE(1, 0, 0);
    V6_STAGE(2, 0, 0);
    V6_STAGE(3, 0, 0);
    V6_STAGE(0, 1, 1);
    V6_STAGE(1, 1, 1);
    V6_STAGE(2, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    V5_BARRIER();
    if (wr == 1) V5_BARRIER();                         // stagger: the wr = 1 group runs one barrier behind

    const int lo0 = frow * 128 + ((fq ^ (frow & 7)) << 4), lo1 = lo0 ^ 64;
    const unsigned char *rdW0 = smem + V5_HALF + wr * (WN * 128) + lo0, *rdW1 = smem + V5_HALF + wr * (WN * 128) + lo1;
    const unsigned char *rdX0 = smem + wc * 4096 + lo0, *rdX1 = smem + wc * 4096 + lo1;
    op16x8 a[4][2], x0[2][2], x1[2][2];

#define V6_LDW(q, buf)                                                                     \
    _Pragma("unroll") for (int blk = 0; blk < FR; blk++) {                                 \
        const int o_ = (buf) * V5_BUF + (q) * 2 * V5_HALF + blk * 2048;                    \
        a[blk][0] = *reinterpret_cast<const op16x8 *>(rdW0 + o_);                          \
        a[blk][1] = *reinterpret_cast<const op16x8 *>(rdW1 + o_);                          \
    }
#define V6_LDX(dst, q, buf)                                                                \
    _Pragma("unroll") for (int blk = 0; blk < 2; blk++) {                                  \
        const int o_ = (buf) * V5_BUF + (q) * 2 * V5_HALF + blk * 2048;                    \
        dst[blk][0] = *reinterpret_cast<const op16x8 *>(rdX0 + o_);                        \
        dst[blk][1] = *reinterpret_cast<const op16x8 *>(rdX1 + o_);                        \
    }
    // quadrant (qa, qb) out: acc[qa*4 + ii][qb*2 + jj][r] = feature qa*128 + wr*64 + ii*16 + 4fq + r, token qb*128 + wc*32 + jj*16 + frow
#define V6_EPI(qa, qb, rsY)                                                                                         \
    if (RES && STG) {     /* fp32 rows through the staging rows: a fragment pair (ii = 2 p, 2 p + 1) of 16 tokens is 16 x 128 B */     \
        _Pragma("unroll") for (int jj = 0; jj < 2; jj++) _Pragma("unroll") for (int p2 = 0; p2 < 2; p2++) {         \
            _Pragma("unroll") for (int e_ = 0; e_ < 2; e_++) {                                                      \
                const int ii = 2 * p2 + e_;                                                                         \
                const unsigned lo_ = CX ? 0u : bb[qa][ii][0], hi_ = CX ? 0u : bb[qa][ii][1];                        \
                const f32x4 bv_ = CX ? (f32x4){0.f, 0.f, 0.f, 0.f} : (f32x4){op16_lo(lo_), op16_hi(lo_), op16_lo(hi_), op16_hi(hi_)};  \
                const f32x4 o4_ = acc[(qa) * 4 + ii][(qb) * 2 + jj] + bv_;                                          \
                V6_STG_WR(__builtin_bit_cast(u32x4, o4_), e_, 0);                                                   \
            }                                                                                                       \
            _Pragma("unroll") for (int it_ = 0; it_ < 2; it_++)                                                     \
                __builtin_amdgcn_raw_buffer_store_b128(V6_STG_RD(it_, 0), rsY, V6_STG_OFF(qb, jj, it_) + ((qa) * HN + p2 * 32) * 4, 0, 0); \
        }                                                                                                           \
    } else if (RES) {                                                                                               \
        _Pragma("unroll") for (int ii = 0; ii < FR; ii++) {                                                         \
            const unsigned lo_ = CX ? 0u : bb[qa][ii][0], hi_ = CX ? 0u : bb[qa][ii][1];                            \
            const f32x4 bv_ = CX ? (f32x4){0.f, 0.f, 0.f, 0.f} : (f32x4){op16_lo(lo_), op16_hi(lo_), op16_lo(hi_), op16_hi(hi_)};  \
            _Pragma("unroll") for (int jj = 0; jj < 2; jj++) {                                                      \
                const f32x4 o4_ = acc[(qa) * 4 + ii][(qb) * 2 + jj] + bv_;                                          \
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4_), rsY,                         \
                                                       V6_VOY((qb) * 2 + jj) + ((qa) * HN + ii * 16) * 4, 0, 0);     \
            }                                                                                                       \
        }                                                                                                           \
    } else {                                                                                                        \
    unsigned c8l_[2][2], c8h_[2][2];      /* c8 rows out: the 8-bit planes of both fragment pairs, stored together */  \
    unsigned c4d_[2][2]; int c4e_[2][2];  /* c4 rows out: packed (lo' x 4 | hi x 4) nibbles and block exponents of both fragment pairs */ \
    _Pragma("unroll") for (int jj = 0; jj < 2; jj++) _Pragma("unroll") for (int pr = 0; pr < 2; pr++) {             \
        if (FR == 3 && pr == 1) {         /* the unpaired third fragment */                                         \
            f32x4 v2_ = acc[(qa) * 4 + 2][(qb) * 2 + jj];                                                           \
            if (EPI == EPI_GELU) { _Pragma("unroll") for (int r = 0; r < 4; r++) v2_[r] = gelu_erf(v2_[r]); }       \
            const op16x2 s0_ = {(op16)v2_[0], (op16)v2_[1]}, s1_ = {(op16)v2_[2], (op16)v2_[3]};                    \
            const u32x2 o2_ = {__builtin_bit_cast(unsigned, s0_), __builtin_bit_cast(unsigned, s1_)};               \
            __builtin_amdgcn_raw_buffer_store_b64(o2_, rsY, V6_VOY((qb) * 2 + jj) + voY3 + (qa) * HN * 2, 0, 0);     \
            continue;                                                                                               \
        }                                                                                                           \
        f32x4 v0_ = acc[(qa) * 4 + 2 * pr][(qb) * 2 + jj], v1_ = acc[(qa) * 4 + 2 * pr + 1][(qb) * 2 + jj];        \
        if (SPLIT == 2) {     /* dual: the raw tile first, to Y */                                                   \
            const op16x2 r0_ = {(op16)v0_[0], (op16)v0_[1]}, r1_ = {(op16)v0_[2], (op16)v0_[3]};                    \
            const op16x2 r2_ = {(op16)v1_[0], (op16)v1_[1]}, r3_ = {(op16)v1_[2], (op16)v1_[3]};                    \
            const auto u0_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, r0_), __builtin_bit_cast(unsigned, r2_), false, false); \
            const auto u1_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, r1_), __builtin_bit_cast(unsigned, r3_), false, false); \
            const u32x4 raw_ = {u0_[0], u1_[0], u0_[1], u1_[1]};                                                    \
            if (STG) V6_STG_WR(raw_, pr, 1);                                                                        \
            else __builtin_amdgcn_raw_buffer_store_b128(raw_, rsY, V6_VOY((qb) * 2 + jj) + ((qa) * HN + pr * 32) * 2, 0, 0); \
        }                                                                                                           \
        if (EPI == EPI_GELU) {                                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                         \
                v0_[r] = gelu_erf(v0_[r]);                                                                          \
                v1_[r] = gelu_erf(v1_[r]);                                                                          \
                if (C8OUT) { v0_[r] = c8_sat(v0_[r]); v1_[r] = c8_sat(v1_[r]); }   /* saturate the VALUE once: its two 8-bit terms then need no clamp (c8.hpp) */ \
                if (SPLIT == 1 || C4OUT) COSA_SPLIT_OPAQUE(v0_[r], v1_[r]);                                         \
            }                                                                                                       \
        }                                                                                                           \
        const op16x2 p0_ = {(op16)v0_[0], (op16)v0_[1]}, p1_ = {(op16)v0_[2], (op16)v0_[3]};                        \
        const op16x2 p2_ = {(op16)v1_[0], (op16)v1_[1]}, p3_ = {(op16)v1_[2], (op16)v1_[3]};                        \
        const auto s0_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, p0_), __builtin_bit_cast(unsigned, p2_), false, false); \
        const auto s1_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, p1_), __builtin_bit_cast(unsigned, p3_), false, false); \
        const u32x4 out_ = {s0_[0], s1_[0], s0_[1], s1_[1]};                                                        \
        if (STG) V6_STG_WR(out_, pr, 0);                                                                            \
        else if (SPLIT == 2) __builtin_amdgcn_raw_buffer_store_b128(out_, (&rsY == &pY) ? pY2 : cY2, V6_VOY((qb) * 2 + jj) + ((qa) * HN + pr * 32) * 2, 0, 0); \
        else __builtin_amdgcn_raw_buffer_store_b128(out_, rsY, V6_VOY((qb) * 2 + jj) + ((qa) * HN + pr * 32) * 2, 0, 0); \
        if (SPLIT == 1) {          /* the lo halves: what the 16-bit rounding above dropped, at column N + n */              \
            const op16x2 q0_ = {(op16)(v0_[0] - (float)p0_[0]), (op16)(v0_[1] - (float)p0_[1])};                    \
            const op16x2 q1_ = {(op16)(v0_[2] - (float)p1_[0]), (op16)(v0_[3] - (float)p1_[1])};                    \
            const op16x2 q2_ = {(op16)(v1_[0] - (float)p2_[0]), (op16)(v1_[1] - (float)p2_[1])};                    \
            const op16x2 q3_ = {(op16)(v1_[2] - (float)p3_[0]), (op16)(v1_[3] - (float)p3_[1])};                    \
            const auto t0_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, q0_), __builtin_bit_cast(unsigned, q2_), false, false); \
            const auto t1_ = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, q1_), __builtin_bit_cast(unsigned, q3_), false, false); \
            const u32x4 lo4_ = {t0_[0], t1_[0], t0_[1], t1_[1]};                                                    \
            if (STG) V6_STG_WR(lo4_, pr, 1);                                                                        \
            else __builtin_amdgcn_raw_buffer_store_b128(lo4_, rsY, V6_VOY((qb) * 2 + jj) + (N + (qa) * HN + pr * 32) * 2, 0, 0); \
        }                                                                                                           \
        if (C8OUT) {       /* c8 rows: lo8 at byte 2N + n, hi8 at byte 3N + n of the row (the window starts at byte 2 n0 of it) */   \
            const float h0_ = (float)p0_[0], h1_ = (float)p0_[1], h2_ = (float)p1_[0], h3_ = (float)p1_[1];        \
            const float h4_ = (float)p2_[0], h5_ = (float)p2_[1], h6_ = (float)p3_[0], h7_ = (float)p3_[1];        \
            const unsigned hA_ = c8_pack4(h0_, h1_, h2_, h3_), hB_ = c8_pack4(h4_, h5_, h6_, h7_);                  \
            const unsigned lA_ = c8_pack4((v0_[0] - h0_) * kC8LoScale, (v0_[1] - h1_) * kC8LoScale, (v0_[2] - h2_) * kC8LoScale, (v0_[3] - h3_) * kC8LoScale); \
            const unsigned lB_ = c8_pack4((v1_[0] - h4_) * kC8LoScale, (v1_[1] - h5_) * kC8LoScale, (v1_[2] - h6_) * kC8LoScale, (v1_[3] - h7_) * kC8LoScale); \
            const auto th_ = __builtin_amdgcn_permlane16_swap(hA_, hB_, false, false);                              \
            const auto tl_ = __builtin_amdgcn_permlane16_swap(lA_, lB_, false, false);                              \
            c8l_[pr][0] = tl_[0]; c8l_[pr][1] = tl_[1]; c8h_[pr][0] = th_[0]; c8h_[pr][1] = th_[1];                 \
            if (pr == 1) {      /* a lane owns 8 consecutive features of each fragment pair; v_permlane32_swap makes that 16 of ONE pair: lanes   \
                                   0-31 take pair 0 (own 8 + those of lane + 32), lanes 32-63 pair 1: one 16-byte store per plane */             \
                const auto l0_ = __builtin_amdgcn_permlane32_swap(c8l_[0][0], c8l_[1][0], false, false);            \
                const auto l1_ = __builtin_amdgcn_permlane32_swap(c8l_[0][1], c8l_[1][1], false, false);            \
                const auto g0_ = __builtin_amdgcn_permlane32_swap(c8h_[0][0], c8h_[1][0], false, false);            \
                const auto g1_ = __builtin_amdgcn_permlane32_swap(c8h_[0][1], c8h_[1][1], false, false);            \
                const int jn0_ = (&rsY == &pY) ? pn0 : n0;                                                          \
                const unsigned v16_ = V6_VOY((qb) * 2 + jj) + fo16m;                                                   \
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){l0_[0], l1_[0], l0_[1], l1_[1]}, rsY, v16_, 2 * N - jn0_ + (qa) * HN, 0); \
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){g0_[0], g1_[0], g0_[1], g1_[1]}, rsY, v16_, 3 * N - jn0_ + (qa) * HN, 0); \
            }                                                                                                       \
        }                                                                                                           \
        if (C4OUT) {       /* c4 rows: the 16 features 16 i + 4 fq + r (fq = 0..3) of fragment i of one token are ONE block, spread over the four   \
                              lanes frow + 16 fq: their maximum meets by lane swaps, every lane converts its four values with the shared scale  \
                              into 16 + 16 bits, and the c8 path's swap network then hands a lane the four dwords of one whole block */          \
            const float hA_[4] = {(float)p0_[0], (float)p0_[1], (float)p1_[0], (float)p1_[1]};                      \
            const float hB_[4] = {(float)p2_[0], (float)p2_[1], (float)p3_[0], (float)p3_[1]};                      \
            float lA_[4], lB_[4], mA_ = 0.f, mB_ = 0.f;                                                              \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                         \
                lA_[r] = (v0_[r] - hA_[r]) * kC4LoScale;                                                            \
                lB_[r] = (v1_[r] - hB_[r]) * kC4LoScale;                                                            \
                mA_ = __builtin_fmaxf(mA_, __builtin_fmaxf(__builtin_fabsf(hA_[r]), __builtin_fabsf(lA_[r])));      \
                mB_ = __builtin_fmaxf(mB_, __builtin_fmaxf(__builtin_fabsf(hB_[r]), __builtin_fabsf(lB_[r])));      \
            }                                                                                                       \
            const int eA_ = c4_block_exp(c4_max4(mA_)), eB_ = c4_block_exp(c4_max4(mB_));                           \
            const float sA_ = c4_pow2(eA_), sB_ = c4_pow2(eB_);                                                     \
            unsigned dA_ = 0, dB_ = 0;                                                                              \
            dA_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dA_, lA_[0], lA_[1], sA_, 0);                            \
            dA_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dA_, lA_[2], lA_[3], sA_, 1);                            \
            dA_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dA_, hA_[0], hA_[1], sA_, 2);                            \
            dA_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dA_, hA_[2], hA_[3], sA_, 3);                            \
            dB_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dB_, lB_[0], lB_[1], sB_, 0);                            \
            dB_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dB_, lB_[2], lB_[3], sB_, 1);                            \
            dB_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dB_, hB_[0], hB_[1], sB_, 2);                            \
            dB_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(dB_, hB_[2], hB_[3], sB_, 3);                            \
            const auto td_ = __builtin_amdgcn_permlane16_swap(dA_, dB_, false, false);                              \
            c4d_[pr][0] = td_[0]; c4d_[pr][1] = td_[1]; c4e_[pr][0] = eA_; c4e_[pr][1] = eB_;                       \
            if (pr == 1) {      /* lanes fq = 0 / 1: block A / B of pair 0, fq = 2 / 3: block A / B of pair 1 (as the c8 stores above) */   \
                const auto x0_ = __builtin_amdgcn_permlane32_swap(c4d_[0][0], c4d_[1][0], false, false);            \
                const auto x1_ = __builtin_amdgcn_permlane32_swap(c4d_[0][1], c4d_[1][1], false, false);            \
                const unsigned g0_ = x0_[0], g1_ = x1_[0], g2_ = x0_[1], g3_ = x1_[1];      /* features 0-3, 4-7, 8-11, 12-15 of the block */ \
                const u32x4 blk_ = {(g0_ & 0xffffu) | (g1_ << 16), (g2_ & 0xffffu) | (g3_ << 16),                    \
                                    (g0_ >> 16) | (g1_ & 0xffff0000u), (g2_ >> 16) | (g3_ & 0xffff0000u)};          \
                const int jn0_ = (&rsY == &pY) ? pn0 : n0, jm0_ = (&rsY == &pY) ? pm0 : m0;                         \
                __builtin_amdgcn_raw_buffer_store_b128(blk_, rsY, V6_VOY((qb) * 2 + jj) + fo16m, 2 * N - jn0_ + (qa) * HN, 0); \
                const int el_ = fq == 0 ? c4e_[0][0] : (fq == 1 ? c4e_[0][1] : (fq == 2 ? c4e_[1][0] : c4e_[1][1])); \
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)c4_scale_byte(el_, 0), sYr, sc_wr_y + ((qb) * 2 + jj) * 2, \
                                                     (((jm0_ >> 8) * (N >> 7) + (jn0_ >> 7) + (qa)) << 11), 0);   \
            }                                                                                                       \
        }                                                                                                           \
        if (STG && pr == 1) {      /* both fragment pairs of these 16 tokens are staged: 8 rows x 128 B per store */  \
            _Pragma("unroll") for (int it_ = 0; it_ < 2; it_++) {                                                   \
                const unsigned vo_ = V6_STG_OFF(qb, jj, it_) + (qa) * HN * 2;                                 \
                if (SPLIT == 2) {      /* dual: plane 1 = the pre-activation (Y), plane 0 = gelu (Y2) */             \
                    __builtin_amdgcn_raw_buffer_store_b128(V6_STG_RD(it_, 1), rsY, vo_, 0, 0);                      \
                    __builtin_amdgcn_raw_buffer_store_b128(V6_STG_RD(it_, 0), (&rsY == &pY) ? pY2 : cY2, vo_, 0, 0); \
                } else {                                                                                            \
                    __builtin_amdgcn_raw_buffer_store_b128(V6_STG_RD(it_, 0), rsY, vo_, 0, 0);                      \
                    if (SPLIT == 1) __builtin_amdgcn_raw_buffer_store_b128(V6_STG_RD(it_, 1), rsY, vo_ + N * 2, 0, 0); \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
    }                                                                                                               \
    }
    // 16 MFMAs of quadrant (qa, qb); FIRST: the accumulation starts from the bias
#define V6_C4ROW(qa, xf, qb, i, ks)                                                                                 \
    do {                                                                                                            \
        const c8_i32x8 a4_ = c4_op(a[i][ks]);                                                                       \
        const int sa_ = (int)swv[(qa) * 2 + ((i) >> 1)], sb_ = (int)sxv[qb];                                        \
        acc[(qa) * 4 + (i)][(qb) * 2] = mfma_c4<((i) & 1) * 2 + (ks), (ks)>(a4_, c4_op(xf[0][ks]), acc[(qa) * 4 + (i)][(qb) * 2], sa_, sb_);             \
        acc[(qa) * 4 + (i)][(qb) * 2 + 1] = mfma_c4<((i) & 1) * 2 + (ks), 2 + (ks)>(a4_, c4_op(xf[1][ks]), acc[(qa) * 4 + (i)][(qb) * 2 + 1], sa_, sb_); \
    } while (0)
#define V6_MMA(qa, xf, qb, FIRST, F8)                                                                               \
    do {                                                                                                            \
        if ((F8) == 2) {      /* fp16c4: two 16x16x128 e2m1 MFMAs per accumulator (k-steps = blocks fq, fq + 4), scale bytes picked by op_sel */ \
            V6_C4ROW(qa, xf, qb, 0, 0); V6_C4ROW(qa, xf, qb, 1, 0); V6_C4ROW(qa, xf, qb, 2, 0); V6_C4ROW(qa, xf, qb, 3, 0);  \
            V6_C4ROW(qa, xf, qb, 0, 1); V6_C4ROW(qa, xf, qb, 1, 1); V6_C4ROW(qa, xf, qb, 2, 1); V6_C4ROW(qa, xf, qb, 3, 1);  \
        } else if (F8) {      /* one 16x16x128 e5m2 MFMA per accumulator: the two 16-byte fragments of a row side by side */ \
            _Pragma("unroll") for (int i = 0; i < FR; i++) {                                                        \
                const c8_i32x8 a8_ = c8_cat(a[i][0], a[i][1]);                                                      \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                       \
                    acc[(qa) * 4 + i][(qb) * 2 + j] = COSA_MFMA_C8(a8_, c8_cat(xf[j][0], xf[j][1]), acc[(qa) * 4 + i][(qb) * 2 + j]); \
            }                                                                                                       \
        } else                                                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                            \
            _Pragma("unroll") for (int i = 0; i < FR; i++) {                                                        \
                f32x4 cb_;                                                                                          \
                if ((FIRST) && ks == 0 && !RES) {                                                                   \
                    const unsigned lo_ = CX ? 0u : bb[qa][i][0], hi_ = CX ? 0u : bb[qa][i][1];      /* c8 / c4: the bias rides in the aug tile */ \
                    cb_[0] = op16_lo(lo_);                                                                          \
                    cb_[1] = op16_hi(lo_);                                                                          \
                    cb_[2] = op16_lo(hi_);                                                                          \
                    cb_[3] = op16_hi(hi_);                                                                          \
                }                                                                                                   \
                _Pragma("unroll") for (int j = 0; j < 2; j++)                                                       \
                    acc[(qa) * 4 + i][(qb) * 2 + j] = COSA_MFMA_16x16x32(                      \
                        a[i][ks], xf[j][ks], ((FIRST) && ks == 0 && !RES) ? cb_ : acc[(qa) * 4 + i][(qb) * 2 + j], 0, 0, 0); \
            }                                                                                                       \
    } while (0)
#define V6_MSECTION_BEGIN()                                    \
    V5_BARRIER();                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_s_setprio(1)
#define V6_MSECTION_END()                                      \
    __builtin_amdgcn_s_setprio(0);                             \
    V5_BARRIER()

    // one K-tile = four phases.  FIRST / LAST are literals; t is the K-tile index inside the job, g the running buffer parity.
    // VMEM operations per wave around a job boundary (st = epilogue stores: 4 op16 / 8 fp32 per quadrant; R' = 8 residual loads):
    //   last.1: DMA(A)        last.2: DMA, st Q00, R' Q00     last.3: DMA, st Q01, R' Q01     last.4: DMA, WAIT(A), st Q11, R' Q11
    //   first.1: DMA(B), WAIT(R' Q00), st Q10, R' Q10, bias     first.2: DMA, WAIT(R' Q01)   first.3: DMA, WAIT(R' Q11)   first.4: DMA, WAIT(B, R' Q10)
    // every WAIT is vmcnt(number of operations issued after its target): op16 out 14 / - / - / - / 6, residual 38 / 38 / 46 / 30 / 6
    // (residual, FR fragments: 6 + 8 FR / 6 + 8 FR / 6 + 10 FR / 6 + 6 FR / 6 -- 30 / 30 / 36 / 24 / 6 for the 192-wide tile)
    // (6 instead of 14 at first.4 also covers a job without predecessor; operations retire in issue order).
#define V6_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define V6_TILE(t, FIRST, LAST, F8)                                                                                 \
    do {                                                                                                            \
        const int b_ = g & 1;                                                                                       \
        /* ---- phase 1 ---- */                                                                                     \
        if ((F8) == 2) { V6_LDSC((t) - Kp - 1); V5_FENCE(); }                                                       \
        V6_LDX(x0, 0, b_);                                                                                          \
        V5_FENCE();                                                                                                 \
        V6_LDW(0, b_);                                                                                              \
        if (!RES && (LAST) && has_next) { V6_LOAD_BIAS(n1); }                                                       \
        V5_FENCE();                                                                                                 \
        if (C4 && !(FIRST) && !(LAST)) {     /* scale chunk c (c4 tiles 2c, 2c + 1) leaves two tiles ahead of its first use */ \
            const int d_ = (t) - (Kp - 1);                                                                          \
            if (d_ >= 0 && !(d_ & 1) && (d_ >> 1) < (Kq >> 1)) V6_SCDMA(d_ >> 1);                                   \
            V5_FENCE();                                                                                             \
        }                                                                                                           \
        V6_STAGE(3, (t) + 1, b_ ^ 1);                                                                               \
        V5_FENCE();                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                          \
        V6_MSECTION_BEGIN();                                                                                        \
        if (RES && (FIRST)) { if (FR == 4) V6_WAIT(38); else V6_WAIT(30); V5_FENCE(); }                             \
        if ((FIRST) && have_prev) {                                                                                 \
            V6_EPI(1, 0, pY);                                                                                       \
            if (RES) { V5_FENCE(); V6_RLOAD(1, 0, cR); V6_LOAD_BIAS(n0); V5_FENCE(); }                              \
        }                                                                                                           \
        V6_MMA(0, x0, 0, FIRST, F8);                                                                                    \
        V6_MSECTION_END();                                                                                          \
        /* ---- phase 2 ---- */                                                                                     \
        V6_LDX(x1, 1, b_);                                                                                          \
        V6_STAGE(0, (t) + 2, b_);                                                                                   \
        V6_MSECTION_BEGIN();                                                                                        \
        if (RES && (FIRST)) { if (FR == 4) V6_WAIT(46); else V6_WAIT(36); V5_FENCE(); }                             \
        if (LAST) {                                                                                                 \
            V6_EPI(0, 0, cY);                                                                                       \
            if (RES && has_next) { V5_FENCE(); V6_RLOAD(0, 0, nR); V5_FENCE(); }                                    \
        }                                                                                                           \
        V6_MMA(0, x1, 1, FIRST, F8);                                                                                    \
        V6_MSECTION_END();                                                                                          \
        /* ---- phase 3 ---- */                                                                                     \
        V6_LDW(1, b_);                                                                                              \
        V6_STAGE(1, (t) + 2, b_);                                                                                   \
        V6_MSECTION_BEGIN();                                                                                        \
        if (RES && (FIRST)) { if (FR == 4) V6_WAIT(30); else V6_WAIT(24); V5_FENCE(); }                             \
        if (LAST) {                                                                                                 \
            V6_EPI(0, 1, cY);                                                                                       \
            if (RES && has_next) { V5_FENCE(); V6_RLOAD(0, 1, nR); V5_FENCE(); }                                    \
        }                                                                                                           \
        V6_MMA(1, x1, 1, FIRST, F8);                                                                                    \
        V6_MSECTION_END();                                                                                          \
        /* ---- phase 4 ---- */                                                                                     \
        V6_STAGE(2, (t) + 2, b_);                                                                                   \
        V5_FENCE();                                                                                                 \
        if ((LAST) && RES) { if (FR == 4) V6_WAIT(38); else V6_WAIT(30); }                                          \
        else if ((LAST) && (C8OUT || C4OUT)) V6_WAIT(22);      /* 6 DMA + 2 x (4 + 4) stores (c4: 4 fp16 + 2 block + 2 scale-byte stores) */ \
        else if (LAST) V6_WAIT(14);                                                                                 \
        else V6_WAIT(6);                                                                                            \
        V6_MSECTION_BEGIN();                                                                                        \
        if (LAST) {                                                                                                 \
            V6_EPI(1, 1, cY);                                                                                       \
            if (RES && has_next) { V5_FENCE(); V6_RLOAD(1, 1, nR); V5_FENCE(); }                                    \
        }                                                                                                           \
        V6_MMA(1, x0, 0, FIRST, F8);                                                                                    \
        V6_MSECTION_END();                                                                                          \
        g++;                                                                                                        \
    } while (0)

    int g = 0;
    while (true) {
        if (C8) {         // Kp + 1 fp16 tiles (hi x hi, aug), then Kp e5m2 tiles: two loop bodies
            V6_TILE(0, 1, 0, 0);
            for (int t = 1; t <= Kp; t++) V6_TILE(t, 0, 0, 0);
            for (int t = Kp + 1; t < nk - 1; t++) V6_TILE(t, 0, 0, 1);
            V6_TILE(nk - 1, 0, 1, 1);
        } else if (C4) {  // Kp + 1 fp16 tiles, then Kq e2m1 tiles
            V6_TILE(0, 1, 0, 0);
            for (int t = 1; t <= Kp; t++) V6_TILE(t, 0, 0, 0);
            for (int t = Kp + 1; t < nk - 1; t++) V6_TILE(t, 0, 0, 2);
            V6_TILE(nk - 1, 0, 1, 2);
        } else {
            V6_TILE(0, 1, 0, 0);
            for (int t = 1; t < nk - 1; t++) V6_TILE(t, 0, 0, 0);
            V6_TILE(nk - 1, 0, 1, 0);
        }
        if (!has_next) break;
        // next job becomes the running one
        pn0 = n0;
        pm0 = m0;
        pY = cY;
        pY2 = cY2;
        have_prev = true;
        o += G;
        m0 = m1;
        n0 = n1;
        cX = nX;
        cW = nW;
        cY = descY(m0, n0);
        if (SPLIT == 2) cY2 = descY2(m0, n0);
        cR = nR;
        has_next = o + G < ntiles_run;
        if (has_next) {
            tile_of(o + G, m1, n1);
            nX = descX(m1);
            nW = descW(n1);
            if (RES) nR = descR(m1, n1);
        } else {
            nX = dead;
            nW = dead;
        }
    }
    V6_EPI(1, 0, cY);
    if (wr == 0) V5_BARRIER();                         // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dead-descriptor DMA of the last K-tiles still writes (zeros) to LDS
    if (stamps && threadIdx.x == 0) atomicMax(&stamps[2 * (blockIdx.x & 63) + 1], __builtin_amdgcn_s_memrealtime());
#undef V6_LOAD_BIAS
#undef V6_RLOAD
#undef V6_WAIT
#undef V6_STAGE
#undef V6_LDW
#undef V6_LDX
#undef V6_EPI
#undef V6_MMA
#undef V6_C4ROW
#undef V6_SCDMA
#undef V6_LDSC
#undef V6_STG_WR
#undef V6_STG_RD
#undef V6_STG_OFF
#undef V6_VOY
#undef V6_MSECTION_BEGIN
#undef V6_MSECTION_END
#undef V6_TILE
}



// =====================================================================================================
// weight gradient  dW[N,K] = dY[M,N]^T X[M,K]   (autograd of nn.Linear; "TN" GEMM, contraction over tokens)
// Both operands have the contraction index as their ROW, so the MFMA fragments (8 consecutive k per lane) are columns of
// the row-major tiles.  The tiles are staged as they are (buffer_load ... lds, rows past M read as zero through the buffer
// bounds check, K-offset in an SGPR so the loop has no address VALU) and the fragments are fetched with gfx950's
// transposing LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major).  LDS image:
// 256-byte rows with the 16-byte chunk index XOR-ed by ((row&3)<<2 | (row>>2)&3) -- conflict-free for these reads -- applied
// on the source side of the DMA.
// Round 3: 256 (features) x 128 (inputs) output tile per 512-thread workgroup (8 waves as 4 x 2, 64 x 64 each), one workgroup per CU,
// THREE 48-KB stages of 64 tokens (two dY images of 128 features + one X image) in a ring: the DMA of stage s + 2 is issued at the start
// of stage s and waited for with a counted vmcnt two stages later (one raw barrier per stage), so a stage's issue -> landed latency
// (~1.2 us) is covered by two stages of MFMA work instead of none -- the round-2 kernel (128 x 128 tiles, two stages, vmcnt(0) per stage,
// two workgroups per CU) spent ~1.7 us per stage of 0.25 us of MFMA work.  The token range is split across workgroups only as far as the
// 256 CUs need it (54 / 18 / 72 / 72 tiles for qkv / proj / fc1 / fc2), and the splits meet WITHOUT atomics: every split writes its fp32
// partial tile to its own slab of a workspace and wgrad_reduce_kernel adds the slabs in split order -- the same bits every run
// (round 2 added the partial tiles with fp32 atomics: run-to-run differences in the last bits, ~1.3 TB/s of atomic traffic).
// =====================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int tr_sw(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// CONV = true: weight gradient of the 3x3 dilated convolution on NHWC tokens (LargeFOV, models/decoder/conv_head.py:11-41):
//   dW9[n][t*Cin + c] = sum_m dY[m][n] * X[src(m, t)][c]        (t = ky*3 + kx, src = the token shifted by the tap, zero outside)
// i.e. the same TN GEMM over an implicit im2col matrix: a k-tile lies inside one tap (Cin % 128 == 0), its X rows are the token rows
// shifted by (dy*w + dx) and rows that leave the image (or M) get an offset past num_records, which the DMA turns into zeros.
// The (image, y, x) position of each staged row is carried from stage to stage (+64 tokens) with adds and compares only.
struct ConvGeom {
    int h, w, dil, cin;          // image height / width in tokens, dilation, channels per tap
    int img_rows, row_off, ldx;  // image b = rows [b*img_rows + row_off, +h*w) of a [*, ldx] matrix
    int q64, r64;                // 64 = q64 * w + r64
};

constexpr int WG_IMG = 16384;                 // one [64 tokens][128 columns] bf16 image
constexpr int WG_STAGE = 3 * WG_IMG;          // dY features [0,128) | dY features [128,256) | X
constexpr int WG_LDS = 3 * WG_STAGE;          // 144 KB: three stages; the fp32 [256][128] epilogue tile (128 KB) reuses them

// one job: the [256 x 128] tile `tile` (n-major) of dW over the token stages [split * stages_per_split, +stages_per_split) -> out (+ split slab)
template <bool CONV>
__device__ __forceinline__ void wgrad_job(unsigned char *smem, const op16 *__restrict__ dY, const op16 *__restrict__ X,
                                          float *__restrict__ out, float *__restrict__ dbp, int M, int N, int K,
                                          int tiles_k, int stages_per_split, int nstages, int tile, int split,
                                          const ConvGeom &cg, int x_bytes, long long slab_elems, int direct_accumulate)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 128;
    const int st0 = split * stages_per_split;
    int st1 = st0 + stages_per_split;
    st1 = st1 < nstages ? st1 : nstages;
    if (st0 >= st1) return;
    const int wr = wave >> 1, wc = wave & 1;          // wave tile: features [wr*64, +64) x inputs [wc*64, +64)
    __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)dY, 0, (int)((size_t)M * N * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, CONV ? x_bytes : (int)((size_t)M * K * 2), 0x00020000);
    // per-lane source offsets of the 2 + 2 + 2 one-KiB pieces this wave stages per stage (fixed for the whole kernel)
    int voY[2][2], voX[2];
    // CONV: position of the staged row (image, y, x) per piece, and the tap of this k-tile
    int pb[2], py[2], px[2], ddy = 0, ddx = 0, cbase = 0;
    if (CONV) {
        const int tap = k0 / cg.cin;
        ddy = (tap / 3 - 1) * cg.dil;
        ddx = (tap % 3 - 1) * cg.dil;
        cbase = k0 - tap * cg.cin;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int q = wave * 2 + i;                  // piece: tile rows 4q .. 4q+3
        const int r = 4 * q + (lane >> 4), ch = (lane & 15) ^ tr_sw(r);
#pragma unroll
        for (int j = 0; j < 2; j++)                  // (a feature half past N: every load out of range -> zeros)
            voY[j][i] = n0 + j * 128 < N ? (r * N + n0 + j * 128 + ch * 8) * 2 : 0x7ffffff0;
        if (CONV) {
            const int tok = st0 * 64 + r, hw = cg.h * cg.w;
            pb[i] = tok / hw;
            const int rem = tok - pb[i] * hw;
            py[i] = rem / cg.w;
            px[i] = rem - py[i] * cg.w;
            voX[i] = (cbase + ch * 8) * 2;           // column part; the row part is added per stage
        } else {
            voX[i] = (r * K + k0 + ch * 8) * 2;
        }
    }
    auto stage = [&](int st, unsigned char *buf) {     // stages are issued in ascending order: the CONV positions advance by 64 tokens per call
        const int soY = st * 64 * N * 2, soX = CONV ? 0 : st * 64 * K * 2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = wave * 2 + i;
            int vx = voX[i];
            if (CONV) {
                const int r = 4 * q + (lane >> 4);
                const int sy = py[i] + ddy, sx = px[i] + ddx;
                const bool in = st * 64 + r < M && sy >= 0 && sy < cg.h && sx >= 0 && sx < cg.w;
                vx = in ? vx + ((pb[i] * cg.img_rows + cg.row_off + sy * cg.w + sx) * cg.ldx) * 2 : 0x7ffffff0;
                px[i] += cg.r64;
                py[i] += cg.q64;
                if (px[i] >= cg.w) { px[i] -= cg.w; py[i]++; }
                while (py[i] >= cg.h) { py[i] -= cg.h; pb[i]++; }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void *)(buf + q * 1024), 16, voY[0][i], soY, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void *)(buf + WG_IMG + q * 1024), 16, voY[1][i], soY, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void *)(buf + 2 * WG_IMG + q * 1024), 16, vx, soX, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient db[n] = sum_m dY[m][n]: the dY^T fragments already hold 8 tokens of one feature per lane, so the workgroups
    // of the first k-tile column (and their wc == 0 waves) add them up on the side
    const bool do_bias = !CONV && dbp != nullptr && tk == 0 && wc == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    // fragment addressing: lane (nn = l&15, g = l>>4) supplies row 8g+4h+q (q = nn>>2), columns 4p..4p+3 (p = nn&3)
    const int nn = lane & 15, g = lane >> 4, q = nn >> 2, p = nn & 3;
    const int ns = st1 - st0;
    stage(st0, smem);
    if (ns > 1) stage(st0 + 1, smem + WG_STAGE);
    for (int s = 0; s < ns; s++) {
        // this wave's pieces of stage s have landed (6 DMA per stage: those of stage s + 1 may stay in flight) ...
        if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everybody's; every wave is also done reading stage s - 1, whose buffer stage s + 2 reuses
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < ns) stage(st0 + s + 2, smem + ((s + 2) % 3) * WG_STAGE);
        // The fragment reads are inline asm: behind a pending LDS-DMA hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
        // ds_read_tr16_b64 BUILTIN (it cannot tell the read from the stage being filled; plain LDS loads do not get that wait), which
        // would drain the two stages in flight every iteration.  The asm reads are ordered by hand: one lgkmcnt(0) per k-step, tied to
        // the fragment registers so that no MFMA can be scheduled above it.
        const unsigned cur = (unsigned)(s % 3) * WG_STAGE;
        const unsigned curA = cur + (wr >> 1) * WG_IMG, curB = cur + 2 * WG_IMG;
        op16x8 a[2][4], b[2][4];
        auto read_frags = [&](int ks) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int r = ks * 32 + 8 * g + 4 * h + q;
                const int sw = tr_sw(r);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int chA = (wr & 1) * 8 + 2 * i + (p >> 1), chB = wc * 8 + 2 * i + (p >> 1);
                    s16x4 va, vb;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(va) : "v"(curA + r * 256 + ((chA ^ sw) << 4) + 8 * (p & 1)));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vb) : "v"(curB + r * 256 + ((chB ^ sw) << 4) + 8 * (p & 1)));
                    reinterpret_cast<s16x4 *>(&a[ks][i])[h] = va;
                    reinterpret_cast<s16x4 *>(&b[ks][i])[h] = vb;
                }
            }
        };
        auto mma = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = COSA_MFMA_16x16x32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int e = 0; e < 8; e++) bsum[i] += (float)a[ks][i][e];
            }
        };
        // the second k-step's fragments are requested before the first one's MFMAs are issued (they land in their shadow)
        read_frags(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]));
        read_frags(1);
        mma(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
        mma(1);
    }
    float *dst = out + (size_t)split * (size_t)slab_elems;
    if (do_bias) {
        // lane (nn, g) holds the partial of feature wr*64 + 16i + nn over its token group g: fold the 4 groups, one store per feature
        float *dbs = dbp + (size_t)split * N;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wr * 64 + 16 * i + nn;
            if (g == 0 && n < N) dbs[n] = direct_accumulate ? dbs[n] + v : v;
        }
    }
    // acc[i][j][r]: n = wr*64 + 16i + 4g + r (row), k' = wc*64 + 16j + nn (col).  Stage the fp32 tile, then whole 512-byte rows.
    __syncthreads();                                      // (every wave is done with the last stage's fragments)
    float *Ct = reinterpret_cast<float *>(smem);          // [256][128] fp32 = 128 KB
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) Ct[(wr * 64 + 16 * i + 4 * g + r) * 128 + wc * 64 + 16 * j + nn] = acc[i][j][r];
    __syncthreads();
    for (int e = tid; e < 256 * 32; e += 512) {
        const int row = e >> 5, c4 = e & 31;
        if (n0 + row < N) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(Ct + row * 128 + 4 * c4);
            float *d = dst + (size_t)(n0 + row) * K + k0 + 4 * c4;
            if (direct_accumulate) v = v + *reinterpret_cast<const f32x4 *>(d);
            *reinterpret_cast<f32x4 *>(d) = v;
        }
    }
}

template <bool CONV>
__global__ __launch_bounds__(512, 1) void gemm_wgrad_kernel(const op16 *__restrict__ dY, const op16 *__restrict__ X,
                                                           float *__restrict__ out, float *__restrict__ dbp, int M, int N, int K,
                                                           int tiles_k, int stages_per_split, int nstages, int ntiles, int tiles_per_xcd,
                                                           ConvGeom cg, int x_bytes, long long slab_elems, int direct_accumulate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Workgroup -> (tile, split).  Workgroups that read the same operand slabs are the tiles of ONE split (the dY slab of an n-tile is
    // shared by its tiles_k k-tiles, the X slab of a k-tile by all n-tiles); ids are dealt round-robin to the 8 XCDs, so give every XCD a
    // contiguous chunk of the n-major tile list (for each split) and its L2 serves the re-reads.
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int split = j / tiles_per_xcd;
    const int tile = xcd * tiles_per_xcd + (j - split * tiles_per_xcd);
    if (tile >= ntiles) return;
    wgrad_job<CONV>(smem, dY, X, out, dbp, M, N, K, tiles_k, stages_per_split, nstages, tile, split, cg, x_bytes, slab_elems, direct_accumulate);
}

// Many weight gradients in ONE persistent launch (the student's encoder: 48 linears, 2592 tiles): with that many tiles no launch needs
// split-K, so every job runs the whole token loop (197 stages at M = 12 560) and writes dW once -- no partial slabs, no reduction
// kernels, and the fixed cost of a launch (ring fill, 128-KB epilogue per workgroup, slab reduction: ~39 of the 53-92 us of a single
// launch) is paid per job of ~190 us instead of per ~15-56 stages.  256 workgroups (one per CU) walk the n-major job list in groups of 32
// consecutive jobs per XCD (workgroup ids are dealt round-robin to the XCDs): the 32 jobs of a group share a few dY / X panels, which
// their L2 then fetches once.
// ---- round 4: 256 x 256 tiles for the batched launch ---------------------------------------------------------------------------------------
// One job = a [256 features] x [256 inputs] tile of dW over ALL tokens; 8 waves as 4 x 2, each 64 features x 128 inputs = 4 x 8 accumulators
// (128 registers, the forward kernel's budget).  Against the 256 x 128 job above a 32-token k-step now feeds 32 MFMAs per wave from 4 LDS-DMA
// instructions and 24 transposing fragment reads (before: 16 MFMAs from 3 and 16) behind one barrier: the token loop's fixed costs per flop
// halve -- that, not operand traffic, is what held the 256 x 128 loop at ~0.45 of the MFMA rate (profiles/r04_wgrad_traffic_probe.txt:
// 12x less distinct operand data bought 5 %).  LDS: a ring of FOUR 32-token k-step buffers of 32 KB ([32][128] images dY0 | dY1 | X0 | X1),
// the DMA runs three k-steps ahead of the reads behind counted vmcnt waits, one raw barrier per k-step.  The products reach every output
// element in the same order as before (32-token k-steps in ascending token order), so the result is bit-identical to the 256 x 128 job's.
constexpr int WW_IMG = 8192;                  // one [32 tokens][128 columns] bf16 image
constexpr int WW_STEP = 4 * WW_IMG;           // dY features [0,128) | [128,256) | X inputs [0,128) | [128,256)
constexpr int WW_LDS = 4 * WW_STEP;           // 128 KB ring; the fp32 epilogue half-tile [128][256] reuses it

__device__ __forceinline__ void wgrad_job_wide(unsigned char *smem, const op16 *__restrict__ dY, const op16 *__restrict__ X,
                                               float *__restrict__ dW, float *__restrict__ db, int M, int N, int K, int tiles_k, int tile)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int nsteps = (M + 31) / 32;
    const int wr = wave >> 1, wc = wave & 1;          // wave tile: features [wr*64, +64) x inputs [wc*128, +128)
    __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)dY, 0, (int)((size_t)M * N * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)((size_t)M * K * 2), 0x00020000);
    // this wave's 1-KiB piece of each image: token rows 4 wave .. 4 wave + 3 of the k-step (fixed per-lane source offsets; halves past N / K: zeros)
    int voY[2], voX[2];
    {
        const int r = 4 * wave + (lane >> 4), ch = (lane & 15) ^ tr_sw(r);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            voY[j] = n0 + j * 128 < N ? (r * N + n0 + j * 128 + ch * 8) * 2 : 0x7ffffff0;
            voX[j] = k0 + j * 128 < K ? (r * K + k0 + j * 128 + ch * 8) * 2 : 0x7ffffff0;
        }
    }
    auto stage = [&](int st) {
        unsigned char *buf = smem + (st & 3) * WW_STEP + wave * 1024;
        const int soY = st * 32 * N * 2, soX = st * 32 * K * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void *)buf, 16, voY[0], soY, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void *)(buf + WW_IMG), 16, voY[1], soY, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void *)(buf + 2 * WW_IMG), 16, voX[0], soX, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void *)(buf + 3 * WW_IMG), 16, voX[1], soX, 0, 0);
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = db != nullptr && tk == 0 && wc == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    // fragment addressing (as in wgrad_job): lane (nn = l & 15, g = l >> 4) supplies token row 8g + 4h + q (q = nn >> 2), columns 4p .. 4p + 3 (p = nn & 3)
    const int nn = lane & 15, g = lane >> 4, q = nn >> 2, p = nn & 3;
    unsigned offA[2][4], offB[2][8];          // byte offsets inside a k-step buffer, per half h of a fragment
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = 8 * g + 4 * h + q, sw = tr_sw(r);
#pragma unroll
        for (int i = 0; i < 4; i++) offA[h][i] = (unsigned)((wr >> 1) * WW_IMG + r * 256 + ((((wr & 1) * 8 + 2 * i + (p >> 1)) ^ sw) << 4) + 8 * (p & 1));
#pragma unroll
        for (int j = 0; j < 8; j++) offB[h][j] = (unsigned)((2 + wc) * WW_IMG + r * 256 + (((2 * j + (p >> 1)) ^ sw) << 4) + 8 * (p & 1));
    }
    stage(0);
    if (nsteps > 1) stage(1);
    if (nsteps > 2) stage(2);
    for (int s = 0; s < nsteps; s++) {
        // this wave's pieces of k-step s have landed (4 DMA per k-step; those of the next two may stay in flight) ...
        if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everybody's; every wave is also done reading k-step s - 1, whose buffer k-step s + 3 reuses
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < nsteps) stage(s + 3);
        const unsigned cur = (unsigned)(s & 3) * WW_STEP;
        op16x8 a[4], b[8];
        // (inline asm reads: behind a pending LDS-DMA hipcc puts vmcnt(0) in front of the ds_read_tr16_b64 builtin -- see wgrad_job)
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s16x4 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(cur + offA[h][i]));
                reinterpret_cast<s16x4 *>(&a[i])[h] = v;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s16x4 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(cur + offB[h][j]));
                reinterpret_cast<s16x4 *>(&b[j])[h] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        // the second half of the X fragments is requested before the first 16 MFMAs are issued (they land in their shadow)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 4; j < 8; j++) {
                s16x4 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(cur + offB[h][j]));
                reinterpret_cast<s16x4 *>(&b[j])[h] = v;
            }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = COSA_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int e = 0; e < 8; e++) bsum[i] += (float)a[i][e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 4; j < 8; j++) acc[i][j] = COSA_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wr * 64 + 16 * i + nn;
            if (g == 0 && n < N) db[n] = v;
        }
    }
    // acc[i][j][r]: feature n = wr*64 + 16i + 4g + r, input k' = wc*128 + 16j + nn.  Two halves of 128 features through a [128][256] fp32 LDS tile,
    // then whole 1-KB rows.
    float *Ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        __syncthreads();                                  // (every wave is done with the last k-step's fragments / the previous half's rows)
        if ((wr >> 1) == hf) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Ct[((wr & 1) * 64 + 16 * i + 4 * g + r) * 256 + wc * 128 + 16 * j + nn] = acc[i][j][r];
        }
        __syncthreads();
        for (int e = tid; e < 128 * 64; e += 512) {
            const int row = e >> 6, c4 = e & 63;
            const int n = n0 + hf * 128 + row, k = k0 + 4 * c4;
            if (n < N && k < K) *reinterpret_cast<f32x4 *>(dW + (size_t)n * K + k) = *reinterpret_cast<const f32x4 *>(Ct + row * 256 + 4 * c4);
        }
    }
}

struct WgradBatchRec { const op16 *dY, *X; float *dW, *db; int N, K, job0, tiles_k; };
constexpr int kWgradBatchMax = 48;
struct WgradBatch { int n, total_jobs, M, pad; WgradBatchRec r[kWgradBatchMax]; };

__global__ __launch_bounds__(512, 1) void gemm_wgrad_batched_kernel(WgradBatch B)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;          // 256 workgroups: 32 per XCD
    for (int it = 0; (it * 8 + xcd) * 32 < B.total_jobs; it++) {
        const int job = (it * 8 + xcd) * 32 + l;
        if (job < B.total_jobs) {                                 // (workgroup-uniform)
            int ri = 0;
            while (ri + 1 < B.n && job >= B.r[ri + 1].job0) ri++;
            const WgradBatchRec &R = B.r[ri];
            wgrad_job_wide(smem, R.dY, R.X, R.dW, R.db, B.M, R.N, R.K, R.tiles_k, job - R.job0);
        }
        __syncthreads();                                          // the next job's first k-step overwrites the epilogue tile
    }
}

// dW = (accumulate ? dW : 0) + slab_0 + slab_1 + ... in split order (and the same for the bias partials): fixed order, no atomics
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int splits, long long elems, float *__restrict__ dW,
                                                          const float *__restrict__ dbp, float *__restrict__ db, int N, int accumulate)
{
    const long long nv = elems >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < nv) {
        f32x4 a = accumulate ? reinterpret_cast<const f32x4 *>(dW)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; s++) a = a + reinterpret_cast<const f32x4 *>(part + (size_t)s * elems)[i];
        reinterpret_cast<f32x4 *>(dW)[i] = a;
    } else if (db != nullptr && i - nv < (N >> 2)) {
        const long long j = i - nv;
        f32x4 a = accumulate ? reinterpret_cast<const f32x4 *>(db)[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; s++) a = a + reinterpret_cast<const f32x4 *>(dbp + (size_t)s * N)[j];
        reinterpret_cast<f32x4 *>(db)[j] = a;
    }
}


// =====================================================================================================
// LargeFOV 3x3 dilated convolution (models/decoder/conv_head.py:11-41: 768->512 and 512->512, dilation 5, padding 5,
// no bias, ReLU) as an implicit GEMM on NHWC tokens:   Y[m, n] = relu( sum_t sum_c X[src(m,t), c] * Wt[t][n][c] ).
// Same 128x128x64 MFMA tile as gemm_bf16_kernel; the K loop runs over 9 taps x Cin/64 chunks.  The activation rows of a
// tap are the token rows shifted by (dy*w + dx); out-of-image taps must contribute zero, which the buffer bounds check
// gives for free: those lanes get an offset past num_records and buffer_load ... lds writes zeros.  The per-lane row
// offsets are recomputed once per tap (9 times per tile), the channel chunk moves through the scalar offset.
// X may be a strided view: image b starts at row b*img_rows + row_off of a [*, ldx] matrix (tokens without the cls row).
// =====================================================================================================
__global__ __launch_bounds__(256, 2) void conv3x3_dil_kernel(const op16 *__restrict__ X, const op16 *__restrict__ Wt,
                                                            op16 *__restrict__ Y, int B, int h, int w, int Cin, int Cout,
                                                            int dil, int img_rows, int row_off, int ldx, int relu,
                                                            long long x_bytes, int tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int M = B * h * w;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int wr = wave >> 1, wc = wave & 1;
    __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)x_bytes, 0x00020000);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the 4 activation rows this lane stages per K-step, decomposed once
    int rb[4], ry[4], rx[4], slot_sw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (wave * 4 + i) + (lane >> 3);
        int m = m0 + row;
        const bool ok = m < M;
        m = ok ? m : 0;
        const int b = m / (h * w), pix = m - b * h * w;
        rb[i] = ok ? b * img_rows + row_off : -1;
        ry[i] = pix / w;
        rx[i] = pix - ry[i] * w;
        slot_sw[i] = ((lane & 7) ^ (row & 7)) * 8;
    }
    int vo[4];
    auto tap_offsets = [&](int t) {
        const int dy = (t / 3 - 1) * dil, dx = (t % 3 - 1) * dil;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int yy = ry[i] + dy, xx = rx[i] + dx;
            const bool ok = rb[i] >= 0 && yy >= 0 && yy < h && xx >= 0 && xx < w;
            vo[i] = ok ? ((rb[i] + yy * w + xx) * ldx + slot_sw[i]) * 2 : 0x7ffffff0;     // past num_records -> zeros
        }
    };
    const int kc = Cin / BK;                 // channel chunks per tap
    const int nk = 9 * kc;
    auto stage = [&](int t, int c0, unsigned char *buf) {
        // weights: plain DMA of Wt[t][n0 .. n0+127][c0 .. c0+63]
        stage_tile(Wt + (size_t)t * Cout * Cin, n0, Cout, Cin, c0, buf, wave, lane);
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void *)(buf + TILE_BYTES + (wave * 4 + i) * 1024), 16, vo[i], c0 * 2, 0, 0);
    };
    tap_offsets(0);
    stage(0, 0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    int tn_ = 0, cn_ = 0;                    // tap / channel offset of the NEXT stage to load
    for (int ks = 0; ks < nk; ks++) {
        unsigned char *cur = smem + (ks & 1) * STAGE_BYTES;
        if (ks + 1 < nk) {
            cn_ += BK;
            if (cn_ == Cin) { cn_ = 0; tn_++; tap_offsets(tn_); }
            stage(tn_, cn_, smem + ((ks + 1) & 1) * STAGE_BYTES);
        }
        const unsigned char *At = cur + (wr * 64) * 128;
        const unsigned char *Bt = cur + TILE_BYTES + (wc * 64) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            op16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = i * 16 + frow;
                const int slot = ((fq + 4 * kk) ^ (row & 7)) << 4;
                a[i] = *reinterpret_cast<const op16x8 *>(At + row * 128 + slot);
                b[i] = *reinterpret_cast<const op16x8 *>(Bt + row * 128 + slot);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = COSA_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned char *Ct = smem;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int nl = wr * 64 + 16 * i + 4 * fq;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ml = wc * 64 + 16 * j + frow;
            op16x4 v;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t = acc[i][j][r];
                if (relu) t = t > 0.f ? t : 0.f;
                v[r] = (op16)t;
            }
            *reinterpret_cast<op16x4 *>(Ct + ml * CT_LD + nl * 2) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = tid; c < BM * 16; c += 256) {
        const int ml = c >> 4, s2 = c & 15;
        if (m0 + ml < M)
            *reinterpret_cast<uint4 *>(Y + (size_t)(m0 + ml) * Cout + n0 + s2 * 8) = *reinterpret_cast<const uint4 *>(Ct + ml * CT_LD + s2 * 16);
    }
}

constexpr size_t kLdsBytes = 128 * (BN * 4 + 16) > 2 * STAGE_BYTES ? 128 * (BN * 4 + 16) : 2 * STAGE_BYTES;

// ---- LayerNorm: fp32 residual stream in, op16 out; one wave per 768-wide row ---------------------------
template <int D>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const op16 *__restrict__ g,
                                                       const op16 *__restrict__ b, op16 *__restrict__ y, float *__restrict__ y32,
                                                       int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int PER = D / 64 / 4;               // float4 per lane
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * D);
    float4 v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) { v[i] = xr[lane + 64 * i]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const float a = v[i].x - mean, c = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
        q += a * a + c * c + d * d + e * e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const op16x4 gg = *reinterpret_cast<const op16x4 *>(g + c0);
        const op16x4 bb = *reinterpret_cast<const op16x4 *>(b + c0);
        float o0 = (v[i].x - mean) * rstd * (float)gg[0] + (float)bb[0];
        float o1 = (v[i].y - mean) * rstd * (float)gg[1] + (float)bb[1];
        float o2 = (v[i].z - mean) * rstd * (float)gg[2] + (float)bb[2];
        float o3 = (v[i].w - mean) * rstd * (float)gg[3] + (float)bb[3];
        if (y) {
            op16x4 ov;
            ov[0] = (op16)o0; ov[1] = (op16)o1; ov[2] = (op16)o2; ov[3] = (op16)o3;
            *reinterpret_cast<op16x4 *>(y + (size_t)row * D + c0) = ov;
        }
        if (y32) *reinterpret_cast<float4 *>(y32 + (size_t)row * D + c0) = make_float4(o0, o1, o2, o3);
    }
}

}  // namespace
}  // namespace cosa

using namespace cosa;

#if COSA_OP_F16        // second build of this file (fp16 operands): the same entry points under their _f16 names (include/cosa_hip.h)
#define cosa_gemm_set_stamp_slot cosa_gemm_set_stamp_slot_f16
#define cosa_gemm_set_variant cosa_gemm_set_variant_f16
#define cosa_gemm_set_grid_policy cosa_gemm_set_grid_policy_f16
#define cosa_gemm_bf16 cosa_gemm_f16
#define cosa_gemm_bf16x3 cosa_gemm_f16x3
#define cosa_layernorm cosa_layernorm_f16
#define cosa_gemm_wgrad_bf16 cosa_gemm_wgrad_f16
#define cosa_conv3x3_dilated_wgrad cosa_conv3x3_dilated_wgrad_f16
#define cosa_conv3x3_dilated_nhwc cosa_conv3x3_dilated_nhwc_f16
#endif

static size_t cosa_c4_scale_bytes_(int rows, int K) { return (size_t)((rows + 255) / 256) * (size_t)(K / 128) * 2048; }
static unsigned long long *g_gemm_stamp_slot = nullptr;
extern "C" void cosa_gemm_set_stamp_slot(void *slot) { g_gemm_stamp_slot = static_cast<unsigned long long *>(slot); }

static int g_gemm_balanced_grid = 0;          // see launch_v6
static int g_gemm_variant = 0;               // 0 = pick per shape (measured, tools/bench_gemm.py); 1 = the 128 x 128 kernel, 6 / 9 = the persistent kernel on 256- / 192-wide jobs (tests)
extern "C" void cosa_gemm_set_variant(int v) { g_gemm_variant = v; }
extern "C" void cosa_gemm_set_grid_policy(int balanced) { g_gemm_balanced_grid = balanced; }

template <int EPI, int SPLIT = 0, int FR = 4>
static int launch_v6(const op16 *x, const op16 *w, const op16 *b, const float *residual, void *Y, int M, int N, int K, hipStream_t st,
                     int ld = 0, int ldy = 0, void *Y2 = nullptr, const unsigned char *xsc = nullptr, const unsigned char *wsc = nullptr,
                     unsigned char *ysc = nullptr)
{
    constexpr size_t lds_bytes = kLdsBytesV5 + (SPLIT == 4 ? kC4ScaleLds + kStageLdsC4 : kStageLds);      // ring [+ fp16c4: scale ring] + output staging rows = 160 KB
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_v6_kernel<EPI, SPLIT, FR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_done = true;
    }
    const int tiles_m = (M + 255) / 256, tiles_n = N / (64 * FR);
    const int ntiles = tiles_m * tiles_n;
    const int grid = ntiles < 256 ? ntiles : 256;          // one persistent workgroup per CU
    // Round quantisation: 256 workgroups walk ntiles jobs, so ntiles = 4 * 256 + 8 (the N = 768 projections of a training step: 1032) costs
    // FIVE rounds, the last with 8 busy CUs.  When the remainder is small, the persistent kernel stops after the full rounds and the
    // leftover 256 x 256 jobs run as 128 x 128 quarters on the two-stage kernel (4 workgroups per job, 2 per CU): a few per cent of a
    // round instead of a whole one.  Same products, same fp32 accumulation order per output element (k ascending) as the jobs it replaces.
    constexpr int tail_max = 48;
    int run = ntiles;
    // ... when it pays: the idle share of the last round must be a sizeable part of the whole launch (> 10 % of its rounds).  The teacher's
    // N = 768 projections (4.03 rounds) qualify; its 12.09- and 16.1-round launches do not -- measured in the step: tail for all three
    // 45.83 / 45.96 ms, for the 4.03-round launches only 45.66 / 45.65, none 46.05 / 46.22
    const int rem = ntiles % 256, rounds_up = (ntiles + 255) / 256;
    // (fp16c8 operands: the 128 x 128 kernel knows their tile sequence for the residual epilogue -- the output projection, N = 768)
    // (three-term operands, SPLIT == 1 -- the default teacher's projections since round 6: the 128 x 128 kernel walks the same 3 K / 64 + 1 tile
    // sequence, so its quarters are interchangeable with the persistent jobs bit for bit; N = 768: 1032 jobs = 4 x 256 + 8)
    constexpr bool can_tail = FR == 4 && (SPLIT == 0 || SPLIT == 1 || (SPLIT == 3 && EPI == EPI_RESIDUAL));
    if (can_tail && ntiles > 256 && rem != 0 && rem <= tail_max && (256 - rem) * 10 > 256 * rounds_up)
        run = ntiles - rem;
    // The persistent grid is balanced over the rounds it needs anyway: 600 jobs are three rounds on 256 workgroups and on 200, and 200
    // leave 56 CUs to whatever else is running (the other stream's kernels, RCCL's channels under DDP: a 256-workgroup launch that finds
    // only 224 free CUs runs its last 32 workgroups AFTER the others -- twice the time).  Multiples of 8 keep a workgroup on one XCD chunk.
    // Single GPU: 0.25 % slower than the full grid (44.57 / 44.74 vs 44.46 / 44.62 ms per step), so it is switched on by the trainer only
    // when the process is one rank of several (cosa_gemm_set_grid_policy).
    const bool balanced = g_gemm_balanced_grid != 0;
    int grid_b = grid;
    if (balanced && run > 256 && M < 40000) {       // (policy 1: the student's launches -- the ones a backward pass overlaps with all-reduces)
        const int rounds = (run + 255) / 256;
        grid_b = ((run + rounds - 1) / rounds + 7) / 8 * 8;
        grid_b = grid_b > 256 ? 256 : grid_b;
    }
    hipLaunchKernelGGL((gemm_bf16_v6_kernel<EPI, SPLIT, FR>), dim3(grid_b), dim3(512), lds_bytes, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n,
                       g_gemm_stamp_slot, ld ? ld : K, ldy ? ldy : N, Y2, run, xsc, wsc, ysc);
    g_gemm_stamp_slot = nullptr;                            // one-shot
    COSA_LAUNCH_CHECK();
    if constexpr (can_tail) {
        if (run < ntiles) {
            // (split 16-bit outputs stage TWO epilogue tiles, hi and lo: 2 x 128 x 272 B, as cosa_gemm_bf16x3 sizes it)
            constexpr int tail_lds = (SPLIT == 1 && 2 * BM * CT_LD > (int)kLdsBytes) ? 2 * BM * CT_LD : (int)kLdsBytes;
            static bool tail_attr = false;
            if (!tail_attr) {
                COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds));
                tail_attr = true;
            }
            hipLaunchKernelGGL((gemm_bf16_kernel<EPI, SPLIT>), dim3(4 * (ntiles - run)), dim3(256), tail_lds, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n,
                               ld ? ld : K, ldy ? ldy : N, static_cast<void *>(nullptr), run);
            COSA_LAUNCH_CHECK();
        }
    }
    return COSA_OK;
}

extern "C" int cosa_gemm_bf16(const void *X, const void *W, const void *bias, const float *residual, void *Y,
                              int M, int N, int K, int epilogue, void *stream)
{
    struct ClearSlot { ~ClearSlot() { g_gemm_stamp_slot = nullptr; } } clear_slot_;      // the stamp slot is one-shot whatever kernel ran
    COSA_REQUIRE(X && W && bias && Y, "cosa_gemm_bf16: null pointer");
    COSA_REQUIRE(M > 0 && N > 0 && K > 0, "cosa_gemm_bf16: bad shape");
    COSA_REQUIRE(N % BN == 0 && K % BK == 0, "cosa_gemm_bf16: N must be a multiple of 128 and K of 64 (got N=%d K=%d)", N, K);
    COSA_REQUIRE(epilogue >= 0 && epilogue <= 2, "cosa_gemm_bf16: unknown epilogue");
    COSA_REQUIRE(epilogue != EPI_RESIDUAL || residual, "cosa_gemm_bf16: residual epilogue needs the residual pointer");
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const dim3 grid(tiles_m * tiles_n), blk(256);
    hipStream_t st = as_stream(stream);
    const op16 *x = static_cast<const op16 *>(X), *w = static_cast<const op16 *>(W), *b = static_cast<const op16 *>(bias);
    // shape rule (measured, profiles/r01_gemm_variants.txt): the persistent 256-wide kernel from 4096 rows up, the 128 x 128 kernel (two
    // workgroups per CU) below that and for operands beyond the 2-GiB reach of its buffer descriptors
    const bool fits_v5 = N % 256 == 0 && M >= 256 && (size_t)(M + 256) * K * 2 < 0x7fffffffull && (size_t)N * K * 2 < 0x7fffffffull;
    constexpr int big_m = 4096;                                  // rows from which the 256 x 256 kernels are used
    const bool fits_v6 = fits_v5 && K >= 128 && (epilogue != EPI_RESIDUAL || g_gemm_variant == 6 || g_gemm_variant == 9 || g_gemm_variant == 0);
    // 192-wide tiles (FR = 3) when they quantise better on the 256 CUs: the student's N = 768 projections (M = 12 560) are 150 jobs of
    // 256 x 256 -- one round at 59 % of the CUs -- but 200 jobs of 256 x 192, one round of 3/4 the length.  Cost model: rounds x job
    // length, the narrow job taken as 0.78 of the wide one (0.75 of the MFMA work, the X panel traffic per MFMA is 4/3).
    if (fits_v6 && N % 192 == 0 && (g_gemm_variant == 9 || (g_gemm_variant == 0 && M >= big_m))) {
        constexpr int wide_max = 48;
        const long tm = (M + 255) / 256, n4 = tm * (N / 256), n3 = tm * (N / 192);
        const double r4 = (n4 > 256 && n4 % 256 != 0 && n4 % 256 <= wide_max) ? (double)(n4 / 256) + 0.1 : (double)((n4 + 255) / 256);
        const double r3 = 0.78 * (double)((n3 + 255) / 256);
        if (g_gemm_variant == 9 || r3 < r4 - 0.05) {
            if (epilogue == EPI_BIAS) return launch_v6<EPI_BIAS, 0, 3>(x, w, b, residual, Y, M, N, K, st);
            if (epilogue == EPI_GELU) return launch_v6<EPI_GELU, 0, 3>(x, w, b, residual, Y, M, N, K, st);
            return launch_v6<EPI_RESIDUAL, 0, 3>(x, w, b, residual, Y, M, N, K, st);
        }
    }
    if (fits_v6 && (g_gemm_variant == 6 || (g_gemm_variant == 0 && M >= big_m))) {       // plain stores
        if (epilogue == EPI_BIAS) return launch_v6<EPI_BIAS>(x, w, b, residual, Y, M, N, K, st);
        if (epilogue == EPI_GELU) return launch_v6<EPI_GELU>(x, w, b, residual, Y, M, N, K, st);
        return launch_v6<EPI_RESIDUAL>(x, w, b, residual, Y, M, N, K, st);
    }
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        attr_done = true;
    }
    switch (epilogue) {
    case EPI_BIAS:
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS>, grid, blk, kLdsBytes, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, K, N, nullptr);
        break;
    case EPI_GELU:
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI_GELU>, grid, blk, kLdsBytes, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, K, N, nullptr);
        break;
    default:
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI_RESIDUAL>, grid, blk, kLdsBytes, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, K, N, nullptr);
        break;
    }
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// (both builds since round 6: the fp16 build exports it as cosa_gemm_f16x3 -- "fp16x3": hi + lo fp16 halves, 11 + 11 significant bits, the
// same three terms at the same cost; tools/sim_precision_map.py scheme `hh`)
// Y = X W^T (+ bias, carried by the augmentation block) with bf16x3 operands -- see split_tile_x / split_tile_w.
//   Xs [M, 2K + 64] = [x_hi | x_lo | 1 1 0 ...],  Ws [N, 2K + 64] = [w_hi | w_lo | b_hi b_lo 0 ...]  (bf16),  zeros: N bf16 zeros
//   epilogue 0 / 1: Y bf16 [M, ldy], columns [0, N) = hi, [N, 2N) = lo of the (GELU'd) result;  epilogue 2: Y fp32 [M, N] = residual + .
extern "C" int cosa_gemm_bf16x3(const void *Xs, const void *Ws, const void *zeros, const float *residual, void *Y,
                                int M, int N, int K, int epilogue, int ldy, void *stream)
{
    struct ClearSlot { ~ClearSlot() { g_gemm_stamp_slot = nullptr; } } clear_slot_;
    COSA_REQUIRE(Xs && Ws && zeros && Y, "cosa_gemm_bf16x3: null pointer");
    COSA_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0, "cosa_gemm_bf16x3: N %% 128 and K %% 64 must be 0 (got N=%d K=%d)", N, K);
    COSA_REQUIRE(epilogue >= 0 && epilogue <= 2, "cosa_gemm_bf16x3: unknown epilogue");
    COSA_REQUIRE(epilogue != EPI_RESIDUAL || residual, "cosa_gemm_bf16x3: residual epilogue needs the residual pointer");
    COSA_REQUIRE(epilogue == EPI_RESIDUAL ? ldy == N : (ldy >= 2 * N && ldy % 8 == 0), "cosa_gemm_bf16x3: ldy must be N (fp32 out) or >= 2N (split out)");
    const int ld = 2 * K + 64;
    hipStream_t st = as_stream(stream);
    const op16 *x = static_cast<const op16 *>(Xs), *w = static_cast<const op16 *>(Ws), *b = static_cast<const op16 *>(zeros);
    if (N % 256 == 0 && M >= 4096 && (size_t)(M + 256) * ld * 2 < 0x7fffffffull * 8 && (size_t)N * ld * 2 < 0x7fffffffull) {
        if (epilogue == EPI_BIAS) return launch_v6<EPI_BIAS, 1>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
        if (epilogue == EPI_GELU) return launch_v6<EPI_GELU, 1>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
        return launch_v6<EPI_RESIDUAL, 1>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
    }
    constexpr int kLdsSplit = 2 * BM * CT_LD > (int)kLdsBytes ? 2 * BM * CT_LD : (int)kLdsBytes;
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_BIAS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsSplit));
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_GELU, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsSplit));
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_RESIDUAL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsSplit));
        attr_done = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const dim3 grid(tiles_m * tiles_n), blk(256);
    switch (epilogue) {
    case EPI_BIAS:
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI_BIAS, 1>), grid, blk, kLdsSplit, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, ld, ldy, nullptr);
        break;
    case EPI_GELU:
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI_GELU, 1>), grid, blk, kLdsSplit, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, ld, ldy, nullptr);
        break;
    default:
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI_RESIDUAL, 1>), grid, blk, kLdsSplit, st, x, w, b, residual, Y, M, N, K, tiles_m, tiles_n, ld, ldy, nullptr);
        break;
    }
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

#if COSA_OP_F16
// Y = X W^T (+ bias, carried by the augmentation block) with fp16c8 operands (c8.hpp; c8_tile_x / c8_tile_w above).
//   Xs [M, 4K + 128 bytes] = [x_hi fp16 | x_lo8 | x_hi8 | 1 1 0 ...],  Ws [N, 4K + 128] = [w_hi | w_lo8 | w_hi8 | b_hi b_lo 0 ...],  zeros: N fp16 zeros
//   epilogue 0: Y fp16 [M, ldy >= N] (plain: the qkv projection feeds the fp16 attention kernel);  1: GELU, Y = c8 rows [M, ldy = 2N + 64
//   fp16 units] (hi | lo8 | hi8; the augmentation block is the consumer's to set);  2: Y fp32 [M, N] = residual + .
extern "C" int cosa_gemm_f16c8(const void *Xs, const void *Ws, const void *zeros, const float *residual, void *Y,
                               int M, int N, int K, int epilogue, int ldy, void *stream)
{
    struct ClearSlot { ~ClearSlot() { g_gemm_stamp_slot = nullptr; } } clear_slot_;
    COSA_REQUIRE(Xs && Ws && zeros && Y, "cosa_gemm_f16c8: null pointer");
    COSA_REQUIRE(M > 0 && N > 0 && K > 0 && N % 256 == 0 && K % 128 == 0, "cosa_gemm_f16c8: N %% 256 == 0 and K %% 128 == 0 required (got M=%d N=%d K=%d)", M, N, K);
    COSA_REQUIRE(epilogue >= 0 && epilogue <= 2, "cosa_gemm_f16c8: unknown epilogue");
    COSA_REQUIRE(epilogue != EPI_RESIDUAL || residual, "cosa_gemm_f16c8: residual epilogue needs the residual pointer");
    COSA_REQUIRE(epilogue == EPI_RESIDUAL ? ldy == N : (epilogue == EPI_GELU ? ldy == 2 * N + 64 : (ldy >= N && ldy % 8 == 0)),
                 "cosa_gemm_f16c8: ldy must be N (fp32 out), 2N + 64 (c8 rows out) or >= N (fp16 out)");
    const int ld = 2 * K + 64;
    COSA_REQUIRE((size_t)256 * ld * 2 < 0x7fffffffull && (size_t)N * ld * 2 < 0x7fffffffull && (size_t)256 * ldy * 4 < 0x7fffffffull, "cosa_gemm_f16c8: panel beyond 2 GiB");
    hipStream_t st = as_stream(stream);
    const op16 *x = static_cast<const op16 *>(Xs), *w = static_cast<const op16 *>(Ws), *b = static_cast<const op16 *>(zeros);
    if (epilogue == EPI_BIAS) return launch_v6<EPI_BIAS, 3>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
    if (epilogue == EPI_GELU) return launch_v6<EPI_GELU, 3>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
    return launch_v6<EPI_RESIDUAL, 3>(x, w, b, residual, Y, M, N, K, st, ld, ldy);
}
#endif

#if COSA_OP_F16
// Y = X W^T (+ bias, carried by the augmentation block) with fp16c4 operands (c4.hpp): rows as for cosa_gemm_f16c8 (same stride; the 8-bit
// planes replaced by the c4 blocks) plus the operands' scale tensors (cosa_c4_scale_bytes; activation / weight layouts).
//   epilogue 0: Y fp16 [M, ldy >= N];  1: GELU, Y = c4 rows [M, ldy = 2N + 64 fp16 units] + Yscales (activation layout, for the next GEMM;
//   the augmentation block is the consumer's to set);  2: Y fp32 [M, N] = residual + .
extern "C" int cosa_gemm_f16c4(const void *Xs, const void *Xscales, const void *Ws, const void *Wscales, const void *zeros, const float *residual,
                               void *Y, void *Yscales, int M, int N, int K, int epilogue, int ldy, void *stream)
{
    struct ClearSlot { ~ClearSlot() { g_gemm_stamp_slot = nullptr; } } clear_slot_;
    COSA_REQUIRE(Xs && Xscales && Ws && Wscales && zeros && Y, "cosa_gemm_f16c4: null pointer");
    COSA_REQUIRE(M > 0 && N > 0 && K >= 256 && N % 256 == 0 && K % 256 == 0, "cosa_gemm_f16c4: N %% 256 == 0 and K %% 256 == 0 required (got M=%d N=%d K=%d)", M, N, K);
    COSA_REQUIRE(epilogue >= 0 && epilogue <= 2, "cosa_gemm_f16c4: unknown epilogue");
    COSA_REQUIRE(epilogue != EPI_RESIDUAL || residual, "cosa_gemm_f16c4: residual epilogue needs the residual pointer");
    COSA_REQUIRE(epilogue != EPI_GELU || Yscales, "cosa_gemm_f16c4: the GELU epilogue writes c4 rows and needs their scale tensor");
    COSA_REQUIRE(epilogue == EPI_RESIDUAL ? ldy == N : (epilogue == EPI_GELU ? ldy == 2 * N + 64 : (ldy >= N && ldy % 8 == 0)),
                 "cosa_gemm_f16c4: ldy must be N (fp32 out), 2N + 64 (c4 rows out) or >= N (fp16 out)");
    const int ld = 2 * K + 64;
    COSA_REQUIRE((size_t)256 * ld * 2 < 0x7fffffffull && (size_t)N * ld * 2 < 0x7fffffffull && (size_t)256 * ldy * 4 < 0x7fffffffull, "cosa_gemm_f16c4: panel beyond 2 GiB");
    COSA_REQUIRE(cosa_c4_scale_bytes_(M, K) < 0x7fffffffull && cosa_c4_scale_bytes_(M, N) < 0x7fffffffull, "cosa_gemm_f16c4: scale tensor beyond 2 GiB");
    hipStream_t st = as_stream(stream);
    const op16 *x = static_cast<const op16 *>(Xs), *w = static_cast<const op16 *>(Ws), *b = static_cast<const op16 *>(zeros);
    const unsigned char *xs = static_cast<const unsigned char *>(Xscales), *ws = static_cast<const unsigned char *>(Wscales);
    unsigned char *ys = static_cast<unsigned char *>(Yscales);
    if (epilogue == EPI_BIAS) return launch_v6<EPI_BIAS, 4>(x, w, b, residual, Y, M, N, K, st, ld, ldy, nullptr, xs, ws, ys);
    if (epilogue == EPI_GELU) return launch_v6<EPI_GELU, 4>(x, w, b, residual, Y, M, N, K, st, ld, ldy, nullptr, xs, ws, ys);
    return launch_v6<EPI_RESIDUAL, 4>(x, w, b, residual, Y, M, N, K, st, ld, ldy, nullptr, xs, ws, ys);
}
#endif

#if !COSA_OP_F16
// training forward of mlp.fc1 (models/vit/vit.py:96-102): H = X W^T + b (bf16, kept for GELU' in the backward) and A = gelu_erf(H) (bf16)
// from ONE pass over the accumulators
extern "C" int cosa_gemm_bf16_dual_gelu(const void *X, const void *W, const void *bias, void *H, void *A, int M, int N, int K, void *stream)
{
    struct ClearSlot { ~ClearSlot() { g_gemm_stamp_slot = nullptr; } } clear_slot_;
    COSA_REQUIRE(X && W && bias && H && A && H != A, "cosa_gemm_bf16_dual_gelu: null pointer / aliased outputs");
    COSA_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0, "cosa_gemm_bf16_dual_gelu: N %% 128 and K %% 64 must be 0 (got N=%d K=%d)", N, K);
    hipStream_t st = as_stream(stream);
    const op16 *x = static_cast<const op16 *>(X), *w = static_cast<const op16 *>(W), *b = static_cast<const op16 *>(bias);
    if (N % 256 == 0 && M >= 4096 && K >= 128 && (size_t)(M + 256) * K * 2 < 0x7fffffffull && (size_t)N * K * 2 < 0x7fffffffull)
        return launch_v6<EPI_GELU, 2>(x, w, b, nullptr, H, M, N, K, st, K, N, A);
    constexpr int kLdsDual = 2 * BM * CT_LD > (int)kLdsBytes ? 2 * BM * CT_LD : (int)kLdsBytes;
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<EPI_GELU, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDual));
        attr_done = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI_GELU, 2>), dim3(tiles_m * tiles_n), dim3(256), kLdsDual, st, x, w, b, static_cast<const float *>(nullptr), H,
                       M, N, K, tiles_m, tiles_n, K, N, A);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
#endif

extern "C" int cosa_layernorm(const float *x, const void *gamma, const void *beta, void *y_bf16, float *y_f32,
                              int rows, int dim, float eps, void *stream)
{
    COSA_REQUIRE(x && gamma && beta && (y_bf16 || y_f32) && rows > 0, "cosa_layernorm: bad arguments");
    COSA_REQUIRE(dim == 768, "cosa_layernorm: dim must be 768 (ViT-B)");
    hipLaunchKernelGGL(layernorm_kernel<768>, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x,
                       static_cast<const op16 *>(gamma), static_cast<const op16 *>(beta), static_cast<op16 *>(y_bf16), y_f32, rows, eps);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// split count of the weight-gradient launch for (M, N, K): one workgroup per CU, never a second, partly filled round; at most 16 slabs
static int wgrad_splits(int M, int N, int K, int *per_out)
{
    const int tiles = ((N + 255) / 256) * (K / 128);
    const int nstages = (M + 63) / 64;
    int splits = tiles >= 256 ? 1 : 256 / tiles;
    if (splits > 16) splits = 16;
    if (splits > nstages) splits = nstages;
    if (splits < 1) splits = 1;
    const int per = (nstages + splits - 1) / splits;
    splits = (nstages + per - 1) / per;
    if (per_out) *per_out = per;
    return splits;
}

#if !COSA_OP_F16
extern "C" size_t cosa_gemm_wgrad_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int splits = wgrad_splits(M, N, K, nullptr);
    return splits > 1 ? align_up((size_t)splits * ((size_t)N * K + N) * sizeof(float), 256) : 256;
}
#endif

template <bool CONV>
static int launch_wgrad(const op16 *dY, const op16 *X, float *dW, float *db, int M, int N, int K, int zero_first, void *workspace,
                        size_t workspace_bytes, hipStream_t st, ConvGeom cg, int x_bytes, const char *who)
{
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_wgrad_kernel<CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS));
        attr_done = true;
    }
    int per = 0;
    const int splits = wgrad_splits(M, N, K, &per);
    const int tiles = ((N + 255) / 256) * (K / 128);
    const int nstages = (M + 63) / 64;
    const size_t elems = (size_t)N * K;
    float *out = dW, *dbp = db;
    if (splits > 1) {
        const size_t need = (size_t)splits * (elems + N) * sizeof(float);
        if (!workspace || workspace_bytes < need) {
            set_error("%s: workspace too small (%zu < %zu): size it with cosa_gemm_wgrad_workspace_bytes", who, workspace_bytes, need);
            return COSA_ENOMEM;
        }
        out = static_cast<float *>(workspace);
        dbp = db ? out + (size_t)splits * elems : nullptr;
    }
    const int tiles_per_xcd = (tiles + 7) / 8;
    hipLaunchKernelGGL(gemm_wgrad_kernel<CONV>, dim3(8 * tiles_per_xcd * splits), dim3(512), WG_LDS, st, dY, X, out, dbp, M, N, K, K / 128, per,
                       nstages, tiles, tiles_per_xcd, cg, x_bytes, (long long)elems, (splits == 1 && !zero_first) ? 1 : 0);
    COSA_LAUNCH_CHECK();
    if (splits > 1) {
        const long long nv = (long long)(elems >> 2) + (db ? (N >> 2) : 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, out, splits, (long long)elems, dW, dbp, db, N,
                           zero_first ? 0 : 1);
        COSA_LAUNCH_CHECK();
    }
    return COSA_OK;
}

extern "C" int cosa_gemm_wgrad_bf16(const void *dY, const void *X, float *dW, float *db, int M, int N, int K, int zero_first,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0, "cosa_gemm_wgrad_bf16: bad arguments");
    COSA_REQUIRE(N % 128 == 0 && K % 128 == 0, "cosa_gemm_wgrad_bf16: N and K must be multiples of 128 (got %d, %d)", N, K);
    COSA_REQUIRE((size_t)M * N * 2 < 0x7fffffffull && (size_t)M * K * 2 < 0x7fffffffull, "cosa_gemm_wgrad_bf16: operand beyond 2 GiB");
    return launch_wgrad<false>(static_cast<const op16 *>(dY), static_cast<const op16 *>(X), dW, db, M, N, K, zero_first, workspace, workspace_bytes,
                               as_stream(stream), ConvGeom{}, 0, "cosa_gemm_wgrad_bf16");
}

#if !COSA_OP_F16
// items: host array of {dY [M,N] bf16, X [M,K] bf16, dW [N,K] f32 (overwritten), db [N] f32 or null (overwritten), N, K}; every item shares M
extern "C" int cosa_gemm_wgrad_batched(const CosaWgradItem *items, int n_items, int M, void *stream)
{
    COSA_REQUIRE(items && n_items > 0 && M > 0, "cosa_gemm_wgrad_batched: bad arguments");
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_wgrad_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS));
        attr_done = true;
    }
    for (int i0 = 0; i0 < n_items; i0 += kWgradBatchMax) {
        WgradBatch B;
        B.n = n_items - i0 < kWgradBatchMax ? n_items - i0 : kWgradBatchMax;
        B.M = M;
        B.pad = 0;
        int jobs = 0;
        for (int i = 0; i < B.n; i++) {
            const CosaWgradItem &it = items[i0 + i];
            COSA_REQUIRE(it.dY && it.X && it.dW && it.N > 0 && it.K > 0, "cosa_gemm_wgrad_batched: bad item %d", i0 + i);
            COSA_REQUIRE(it.N % 128 == 0 && it.K % 128 == 0, "cosa_gemm_wgrad_batched: N and K must be multiples of 128 (item %d: %d, %d)", i0 + i, it.N, it.K);
            COSA_REQUIRE((size_t)M * it.N * 2 < 0x7fffffffull && (size_t)M * it.K * 2 < 0x7fffffffull, "cosa_gemm_wgrad_batched: operand beyond 2 GiB");
            B.r[i] = WgradBatchRec{static_cast<const op16 *>(it.dY), static_cast<const op16 *>(it.X), it.dW, it.db, it.N, it.K, jobs, (it.K + 255) / 256};
            jobs += ((it.N + 255) / 256) * ((it.K + 255) / 256);
        }
        for (int i = B.n; i < kWgradBatchMax; i++) B.r[i] = WgradBatchRec{nullptr, nullptr, nullptr, nullptr, 0, 0, jobs, 1};
        B.total_jobs = jobs;
        hipLaunchKernelGGL(gemm_wgrad_batched_kernel, dim3(256), dim3(512), WW_LDS, as_stream(stream), B);
        COSA_LAUNCH_CHECK();
    }
    return COSA_OK;
}
#endif

// weight gradient of cosa_conv3x3_dilated_nhwc:  dW9[Cout][9*Cin] (fp32, tap-major columns: t*Cin + c) = (zero_first ? 0 : dW9) +
// dY[B*h*w, Cout]^T im2col(X); X is addressed exactly as in the forward call (strided token view, image b at row b*img_rows + row_off).
// workspace: cosa_gemm_wgrad_workspace_bytes(B*h*w, Cout, 9*Cin)
extern "C" int cosa_conv3x3_dilated_wgrad(const void *dY, const void *X, float *dW9, int B, int h, int w, int Cin, int Cout, int dilation,
                                          int img_rows, int row_off, int ldx, int zero_first, void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(dY && X && dW9 && B > 0 && h > 0 && w > 0 && dilation > 0, "cosa_conv3x3_dilated_wgrad: bad arguments");
    COSA_REQUIRE(Cin % 128 == 0 && Cout % 128 == 0, "cosa_conv3x3_dilated_wgrad: Cin and Cout must be multiples of 128 (got %d, %d)", Cin, Cout);
    COSA_REQUIRE(img_rows >= h * w + row_off && ldx >= Cin, "cosa_conv3x3_dilated_wgrad: bad view geometry");
    const long long x_bytes = (long long)B * img_rows * ldx * 2;
    const int M = B * h * w, N = Cout, K = 9 * Cin;
    COSA_REQUIRE(x_bytes < 0x7fffff00ll && (size_t)M * N * 2 < 0x7fffffffull, "cosa_conv3x3_dilated_wgrad: operand beyond 2 GiB");
    ConvGeom cg{h, w, dilation, Cin, img_rows, row_off, ldx, 64 / w, 64 % w};
    return launch_wgrad<true>(static_cast<const op16 *>(dY), static_cast<const op16 *>(X), dW9, nullptr, M, N, K, zero_first, workspace,
                              workspace_bytes, as_stream(stream), cg, (int)x_bytes, "cosa_conv3x3_dilated_wgrad");
}

extern "C" int cosa_conv3x3_dilated_nhwc(const void *X, const void *Wt, void *Y, int B, int h, int w, int Cin, int Cout, int dilation,
                                         int img_rows, int row_off, int ldx, int relu, void *stream)
{
    COSA_REQUIRE(X && Wt && Y && B > 0 && h > 0 && w > 0 && dilation > 0, "cosa_conv3x3_dilated_nhwc: bad arguments");
    COSA_REQUIRE(Cin % 64 == 0 && Cout % 128 == 0, "cosa_conv3x3_dilated_nhwc: Cin %% 64 and Cout %% 128 must be 0");
    COSA_REQUIRE(img_rows >= h * w + row_off && ldx >= Cin, "cosa_conv3x3_dilated_nhwc: bad view geometry");
    const long long x_bytes = (long long)B * img_rows * ldx * 2;
    COSA_REQUIRE(x_bytes < 0x7ffffff0ll, "cosa_conv3x3_dilated_nhwc: activation view beyond 2 GiB");
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)conv3x3_dil_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        attr_done = true;
    }
    const int M = B * h * w;
    const int tiles_m = (M + 127) / 128, tiles_n = Cout / 128;
    hipLaunchKernelGGL(conv3x3_dil_kernel, dim3(tiles_m * tiles_n), dim3(256), kLdsBytes, as_stream(stream), static_cast<const op16 *>(X),
                       static_cast<const op16 *>(Wt), static_cast<op16 *>(Y), B, h, w, Cin, Cout, dilation, img_rows, row_off, ldx, relu,
                       x_bytes, tiles_n);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
