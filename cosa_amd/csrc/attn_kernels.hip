// attn_kernels.hip -- ViT patch attention for gfx950 (CDNA4): LDS-tiled, MFMA, online softmax.
//
// Reference: models/vit/vit.py:119-137 (Attention.forward): softmax(q k^T * hd^-0.5) v, 12 heads x 64,
// sequence lengths 197 / 785 / 1765 (+3601 at 640 crops).  The reference materialises the
// [B,12,N,N] score tensor; here it never leaves the CU.
//
// Layouts (op16):  qkv [B, N, 3, H, 64]  (straight out of the qkv projection, no permute copy)
//                  (V is read in place: the PV product fetches its V^T fragments with the transposing LDS read, no V^T copy)
//                  out [B, N, H*64]      (what the output projection consumes)
//                  lse [B, H, N] f32     natural-log sum-exp of the scaled scores (for backward)
//
// Wave64 tiling: a workgroup = 4 waves = 128 queries; each wave owns 32 queries and walks the keys
// in tiles of 64.  Scores are computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that
// the query sits on the lane: row max / row sum are in-lane plus one cross-half exchange, the
// exponentiated tile is already the B operand of O^T = V^T P^T (no LDS round trip, no shuffles),
// and the O rescale / final 1/l are lane-local.
#include "kernels.hpp"
#include "op16.hpp"
#include "c8.hpp"
#include "c4.hpp"
#include <type_traits>
#include <cstdlib>

namespace cosa {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;           // head dim
constexpr int BQ = 128;          // queries per workgroup
constexpr int BK = 64;           // keys per tile

// forward V tile: row-major [64 keys][64 d] (128-B rows, exactly as it lies in qkv), read as the A operand of O^T = V^T P^T with the
// transposing LDS read ds_read_b64_tr_b16 (a 16-lane group fetches a 4-key x 16-d block, lane nn gets column nn's 4 keys): no V^T
// copy in HBM, no transpose kernel.  16-B chunk c of row r sits at chunk c ^ (((r >> 1) & 1) << 2): the 4 rows x 64 B a 32-lane half
// reads then cover all 64 banks once (rows r, r+1 land 128 B apart = 32 banks; the XOR moves rows r+2, r+3 by 16 banks).
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4v lds_s16x4v;
typedef __attribute__((address_space(3))) void lds_void_a;
__device__ __forceinline__ int vsw(int row) { return ((row >> 1) & 1) << 2; }
__device__ __forceinline__ op16x8 v_frag(const unsigned char *Vs, int keyb, int dhalf, int lane)
{
    const int nn = lane & 15, grp = (lane >> 4) & 1;
    const int r0 = keyb + (nn >> 2), r1 = r0 + 8;
    const int ch = dhalf * 4 + grp * 2 + ((nn & 3) >> 1), off = 8 * (nn & 1);
    union { s16x4v h[2]; op16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4v *)(Vs + r0 * 128 + ((ch ^ vsw(r0)) << 4) + off));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4v *)(Vs + r1 * 128 + ((ch ^ vsw(r1)) << 4) + off));
    return u.v;
}

constexpr float kDeferLog2 = 6.0f;
// max of three floats: one v_max3_f32 (this file is built with -fno-honor-nans: no canonicalising v_max x, x per input in front of fmaxf).
// NOT inline asm: its operands are MFMA accumulators, and the wait states between an MFMA and a VALU read of its result are inserted by the
// compiler's hazard recognizer, which does not look into asm statements.  Rounds 2-4 had `asm("v_max3_f32 ...")` here: correct only as long as
// the schedule kept the statement >= 64 cycles behind the producing MFMA (it did: the builds are bit-identical to this one), and run-to-run
// NON-deterministic as soon as the surrounding s_setprio fences were removed (round 5: tools/ab_attn_libs.py, profiles/r05_attn_variants.txt).
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// c + p[0] + p[1] of a packed pair of probabilities, fp32 accumulation (v_dot2c_f32_f16 / _bf16 with the constant (1, 1))
__device__ __forceinline__ float pair_sum(op16x2 p, float c)
{
#if COSA_OP_F16
    return __builtin_amdgcn_fdot2(p, (op16x2){(op16)1.f, (op16)1.f}, c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(p, (op16x2){(op16)1.f, (op16)1.f}, c, false);
#endif
}
__device__ __forceinline__ int crow(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// Workgroup -> (query/key block, batch, head).  Workgroups are dealt round-robin to the 8 XCDs (id & 7), each with a private L2;
// all blocks of one (batch, head) therefore take consecutive slots OF ONE XCD, so its K/V (or Q/dO) panel is fetched into
// one L2 once instead of into all eight (the x-fastest 3-D grid spread them: 1.65 GB of HBM reads for 0.43 GB of operands).
__device__ __forceinline__ bool attn_block_map(int nblk, int ngroups, int H, int &blk, int &b, int &h)
{
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int gl = j / nblk;
    blk = j - gl * nblk;
    const int grp = gl * 8 + xcd;
    if (grp >= ngroups) return false;
    b = grp / H;
    h = grp - b * H;
    return true;
}

// ---- forward -----------------------------------------------------------------------------------------
// One pass over the keys in 64-key tiles with the online softmax; the query sits on the LANE (r = lane & 31; the two 32-lane halves hh
// own interleaved 4-row groups of the MFMA output, crow()):
//   * S^T[key][query] = K Q^T for the tile's two 32-key blocks (v_mfma_f32_32x32x16, A = K rows from LDS, B = the Q fragments held in
//     registers for the whole kernel);
//   * raw-score maximum per query over the lane's registers and the other half-lane (one cross-lane op), the scale folded into the exp2
//     argument (one fma per score); the running O and sum are rescaled only when some row's maximum moved (wave-uniform test);
//   * O^T[d][query] += V^T[d][key] P^T[key][query]: the exponentiated accumulators, rounded to the operand type, ARE the B operand of the
//     second MFMA (no LDS round trip for P), V^T fragments come from the transposing LDS read (v_frag above);
//   * at the end the two half-lane partial sums are combined, O is normalised and stored, LSE = (m + log2 l) ln 2.
// stamps: optional device-side span of the launch (100 MHz wall clock; min start / max end over workgroups) -- HIP events cannot be
// recorded inside a captured hipGraph on ROCm, so bench.py reads these instead.

// ---- forward with bf16x3 ("split") operands: the parity-grade no-grad passes ------------------------------------------------------
// (both builds since round 6: with fp16 operands this is the attention of the "fp16x3" mode -- hi + lo fp16 halves.  The un-normalised
// probabilities are <= 1 with the row maximum at 1: a lo half that falls into fp16's subnormal range belongs to a probability below 2^-3 of
// the maximum and carries an absolute 2^-25 of it -- nothing is scaled.  The fp16 MFMA takes subnormal operands as they are (measured).)
// The algorithm above, 4 waves x 32 queries per workgroup; q, k, v and the probabilities are carried as hi + lo bf16 halves (16 significant
// bits) and every product is the three MFMA terms hi*hi + hi*lo + lo*hi with fp32 accumulation:
//     S^T  = K_h Q_h^T + K_h Q_l^T + K_l Q_h^T          O^T += V_h^T P_h^T + V_h^T P_l^T + V_l^T P_h^T
// qkv rows are the split output of the qkv projection: [hi (3*H*64) | lo (3*H*64)], row stride ldq; the output rows are split rows for
// the output projection: [hi (H*64) | lo (H*64) | aug (1, 1, 0, ...)], row stride ldo (gemm_kernels.hip: split_tile_x / split_tile_w).
// K / V tiles (hi and lo halves: four [64][64] images, 32 KB) arrive by LDS-DMA into a 2-deep ring -- the next tile is in flight while this
// one is computed, one barrier per tile, no staging registers.  (The first version staged them through 32 registers that the compiler
// kept in scratch: every tile waited for its global loads to store them, 1.02 ms per launch of the teacher's mix against 0.14 ms for
// the bf16 kernel.)
__global__ __launch_bounds__(256) void attn_fwd_x3_kernel(const op16 *__restrict__ qkv, op16 *__restrict__ out, float *__restrict__ lse,
                                                         int N, int H, int nblk, int ngroups, float scale_log2e, int ldq, int ldo,
                                                         unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // [2 stages][Kh | Kl | Vh | Vl][64 * 128]
    if (stamps && threadIdx.x == 0) atomicMin(&stamps[2 * (blockIdx.x & 63)], __builtin_amdgcn_s_memrealtime());     // device-clock span of the launch, as attn_fwd2_kernel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    int blk, b, h;
    if (!attn_block_map(nblk, ngroups, H, blk, b, h)) return;
    const int q0 = blk * BQ + wave * 32;
    const size_t rs = (size_t)ldq;
    const size_t lo_off = (size_t)3 * H * HD;

    const int qrow = min(q0 + r, N - 1);
    const op16 *qp = qkv + ((size_t)b * N + qrow) * rs + h * HD + 8 * hh;
    op16x8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        qh[s] = *reinterpret_cast<const op16x8 *>(qp + 16 * s);
        ql[s] = *reinterpret_cast<const op16x8 *>(qp + lo_off + 16 * s);
    }
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; i++) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -INFINITY, l = 0.f;

    const op16 *kbase = qkv + (size_t)b * N * rs + (size_t)H * HD + h * HD;
    const op16 *vbase = kbase + (size_t)H * HD;
    const int swz = (r >> 1) & 7;
    // DMA: wave w fills the 1-KiB pieces 2w, 2w + 1 (8 rows each) of each of the four images; swizzles on the source side, rows past
    // the last key read as zeros through the buffer range check
    const int nbytes = (int)(((size_t)(N - 1) * rs + HD) * 2);
    const __amdgpu_buffer_rsrc_t rsKh = __builtin_amdgcn_make_buffer_rsrc((void *)kbase, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsKl = __builtin_amdgcn_make_buffer_rsrc((void *)(kbase + lo_off), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVh = __builtin_amdgcn_make_buffer_rsrc((void *)vbase, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVl = __builtin_amdgcn_make_buffer_rsrc((void *)(vbase + lo_off), 0, nbytes, 0x00020000);
    unsigned voK[2], voV[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (2 * wave + i) + (lane >> 3), ps = lane & 7;
        voK[i] = (unsigned)((row * rs + (ps ^ ((row >> 1) & 7)) * 8) * 2);
        voV[i] = (unsigned)((row * rs + (ps ^ vsw(row)) * 8) * 2);
    }
    auto dma_tile = [&](int k0, int buf) {
        const unsigned ko = (unsigned)((size_t)k0 * rs * 2);
        unsigned char *d = smem + buf * 4 * BK * 128 + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsKh, (lds_void_a *)(d + i * 1024), 16, voK[i] + ko, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsKl, (lds_void_a *)(d + BK * 128 + i * 1024), 16, voK[i] + ko, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsVh, (lds_void_a *)(d + 2 * BK * 128 + i * 1024), 16, voV[i] + ko, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsVl, (lds_void_a *)(d + 3 * BK * 128 + i * 1024), 16, voV[i] + ko, 0, 0, 0);
        }
    };
    int ring = 0;
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float NEG_INF = -INFINITY;
    auto tile = [&](int k0, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        const unsigned char *Kh = smem + ring * 4 * BK * 128, *Kl = Kh + BK * 128, *Vh = Kh + 2 * BK * 128, *Vl = Kh + 3 * BK * 128;
        if (k0 + BK < N) dma_tile(k0 + BK, ring ^ 1);

        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; i++) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int slot = ((2 * s + hh) ^ swz) << 4;
            const op16x8 a0h = *reinterpret_cast<const op16x8 *>(Kh + r * 128 + slot);
            const op16x8 a1h = *reinterpret_cast<const op16x8 *>(Kh + (r + 32) * 128 + slot);
            const op16x8 a0l = *reinterpret_cast<const op16x8 *>(Kl + r * 128 + slot);
            const op16x8 a1l = *reinterpret_cast<const op16x8 *>(Kl + (r + 32) * 128 + slot);
            s0 = COSA_MFMA_32x32x16(a0l, qh[s], s0, 0, 0, 0);          // small terms first
            s1 = COSA_MFMA_32x32x16(a1l, qh[s], s1, 0, 0, 0);
            s0 = COSA_MFMA_32x32x16(a0h, ql[s], s0, 0, 0, 0);
            s1 = COSA_MFMA_32x32x16(a1h, ql[s], s1, 0, 0, 0);
            s0 = COSA_MFMA_32x32x16(a0h, qh[s], s0, 0, 0, 0);
            s1 = COSA_MFMA_32x32x16(a1h, qh[s], s1, 0, 0, 0);
        }
        if constexpr (tail) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (k0 + crow(i, hh) >= N) s0[i] = NEG_INF;
                if (k0 + 32 + crow(i, hh) >= N) s1[i] = NEG_INF;
            }
        }
        float mt = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; i++) mt = fmaxf(mt, fmaxf(s0[i], s1[i]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mnew = fmaxf(m, mt * scale_log2e);
        if (__any(mnew != m)) {
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; }
            m = mnew;
        }
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            s0[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[i], scale_log2e, -m));
            s1[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], scale_log2e, -m));
            ls += s0[i] + s1[i];
        }
        l += ls;
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int sp = 0; sp < 2; sp++) {
                op16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float p = kb == 0 ? s0[8 * sp + j] : s1[8 * sp + j];
                    ph[j] = (op16)p;
                    pl[j] = (op16)(p - (float)ph[j]);
                }
                const int keyb = kb * 32 + 16 * sp + 4 * hh;
                const op16x8 vh0 = v_frag(Vh, keyb, 0, lane), vh1 = v_frag(Vh, keyb, 1, lane);
                const op16x8 vl0 = v_frag(Vl, keyb, 0, lane), vl1 = v_frag(Vl, keyb, 1, lane);
                o0 = COSA_MFMA_32x32x16(vl0, ph, o0, 0, 0, 0);
                o1 = COSA_MFMA_32x32x16(vl1, ph, o1, 0, 0, 0);
                o0 = COSA_MFMA_32x32x16(vh0, pl, o0, 0, 0, 0);
                o1 = COSA_MFMA_32x32x16(vh1, pl, o1, 0, 0, 0);
                o0 = COSA_MFMA_32x32x16(vh0, ph, o0, 0, 0, 0);
                o1 = COSA_MFMA_32x32x16(vh1, ph, o1, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next tile has landed and nobody still reads this one
        __syncthreads();
        ring ^= 1;
    };
    const int nfull = (N / BK) * BK;
    for (int k0 = 0; k0 < nfull; k0 += BK) tile(k0, std::false_type{});
    if (nfull < N) tile(nfull, std::true_type{});
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int q = q0 + r;
    if (q < N) {
        op16 *op = out + ((size_t)b * N + q) * ldo + h * HD;
        const int lo_o = H * HD;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            op16x4 h0, h1, l0, l1;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // The products must be ROUNDED fp32 values before they are split: under -ffp-contract=fast the fp16 build fused the multiply
                // into one of the two conversions (v_fma_mixlo_f16: one rounding of the exact product) and not into the other (v_mul_f32 +
                // v_cvt: two roundings), so that on a double-rounding tie -- 2 to 4 elements in 150 000 -- the stored hi half and the hi half
                // the lo half was formed from differed by one fp16 ulp (round 6, found by the fp16x3 kernel test).
                // The empty asm makes each product an opaque fp32 register (no instruction is emitted; the backend's fusion is a global
                // option under -ffp-contract=fast and ignores `#pragma clang fp contract(off)`).
                float a = o0[4 * g + j] * inv, c = o1[4 * g + j] * inv;
                asm volatile("" : "+v"(a), "+v"(c));
                h0[j] = (op16)a; l0[j] = (op16)(a - (float)h0[j]);
                h1[j] = (op16)c; l1[j] = (op16)(c - (float)h1[j]);
            }
            *reinterpret_cast<op16x4 *>(op + 8 * g + 4 * hh) = h0;
            *reinterpret_cast<op16x4 *>(op + 32 + 8 * g + 4 * hh) = h1;
            *reinterpret_cast<op16x4 *>(op + lo_o + 8 * g + 4 * hh) = l0;
            *reinterpret_cast<op16x4 *>(op + lo_o + 32 + 8 * g + 4 * hh) = l1;
        }
        if (h == 0) {                                   // augmentation block of this token row: (1, 1, 0, ...), 64 B per half-lane
            op16 *ap = out + ((size_t)b * N + q) * ldo + 2 * lo_o + 32 * hh;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                op16x8 a;
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = (op16)0.f;
                if (hh == 0 && c == 0) { a[0] = (op16)1.f; a[1] = (op16)1.f; }
                *reinterpret_cast<op16x8 *>(ap + 8 * c) = a;
            }
        }
        if (hh == 0 && lse) lse[((size_t)b * H + h) * N + q] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    }
    if (stamps) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&stamps[2 * (blockIdx.x & 63) + 1], __builtin_amdgcn_s_memrealtime());
    }
}

// ---- forward, 64 queries per wave ---------------------------------------------------------------------------------
// The algorithm above; a workgroup is 2 waves, each wave owns TWO 32-query
// blocks.  Every K / V^T fragment read from LDS feeds two MFMAs (half the LDS fragment traffic per flop of a 32-query wave), each wave
// has two independent MFMA/softmax chains to interleave, and at ~200 VGPRs four such workgroups share a CU, so one
// workgroup's barrier / staging phase overlaps the others' compute.
// DMA: K/V tiles arrive by LDS-DMA (buffer_load ... lds) into a 2-deep ring -- the next tile is in flight while this one is
// computed, one barrier per tile, no staging registers; the swizzles are applied on the source side (lane-linear LDS image) and
// key rows past N read as zeros through the buffer range check.
// C8OUT (fp16 build): the output leaves as fp16c8 rows (c8.hpp: hi fp16 | lo8 | hi8 | aug (1, 1, 0, ...), row stride 4 H HD + 128 bytes)
// for the c8 output projection: q, k, v and P stay plain fp16 (the pseudo labels are insensitive to 11-bit attention operands, but not to
// an 11-bit attention OUTPUT: at near-uniform attention the output is a large common mean plus a small token-specific part that the
// projection must still see -- tools/sim_precision_map.py)
// NW: waves per workgroup (64 queries each) sharing one K/V tile stream.  Every LDS-DMA instruction costs its wave 100-185 issue cycles
// (timing ablation at N = 1765: the DMA issue was a quarter of the kernel) and a tile is 16 of them whatever the workgroup size: with
// NW = 4 a wave issues 4 per tile instead of 8.  Long sequences take NW = 4 (two workgroups per CU), short ones NW = 2 (finer query blocks).
// OUTM = 2 (fp16 build): the output leaves as fp16c4 rows (c4.hpp): the 16 columns 32 d + 16 gp .. + 15 of a query are one MX block, held by
// the query's two lanes (hh = 0 / 1): their maxima meet by one lane swap, every lane converts its eight values with the shared scale, a
// second swap hands the hh = 0 lane the block's lo' half and the hh = 1 lane its hi half (8-byte stores), the hh = 0 lane stores the scale
// byte at the row's place in the operand's scale tensor (row0 = first row of this launch's images in that operand).
// AUGM (round 4): the softmax's "scale and subtract the running maximum" leaves the VALU.  Q is held pre-multiplied by scale * log2(e), and
// the running reference m[u] of a query enters the score MFMAs as a 65th contraction index (K side: 1 for every key, Q side: -m, a register
// constant per query that changes only when the reference moves): the accumulators come out as  s * scale * log2 e - m, ready for v_exp.
// m is kept representable in the operand type, so the same value is subtracted everywhere (the factor 2^m cancels between O and l).  One
// more MFMA per 32 x 32 score block (4 on 16) for 64 fewer VALU fmas per lane and tile: the kernel is VALU-issue bound.
template <bool DMA, int OUTM = 0, int NW = 2, bool AUGM = false>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const op16 *__restrict__ qkv, const op16 *__restrict__ vt,
                                                       op16 *__restrict__ out, float *__restrict__ lse,
                                                       int N, int Npad, int H, int nblk, int ngroups, float scale_log2e,
                                                       unsigned long long *__restrict__ stamps, unsigned char *__restrict__ out_scales = nullptr, int row0 = 0)
{
    constexpr bool C8OUT = OUTM == 1, C4OUT = OUTM == 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(DMA ? 4 : 2) * BK * 128];
    if (stamps && threadIdx.x == 0) atomicMin(&stamps[2 * (blockIdx.x & 63)], __builtin_amdgcn_s_memrealtime());     // 64 shards: one address would serialise the workgroups' atomics
    unsigned char *Ks = smem;
    unsigned char *Vs = smem + BK * 128;
    (void)vt;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    int blk, b, h;
    if (!attn_block_map(nblk, ngroups, H, blk, b, h)) return;
    const int q0 = blk * (64 * NW) + wave * 64;
    const bool busy = q0 < N;          // wave-uniform
    const size_t rs = (size_t)3 * H * HD;
    op16x8 qf[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int qrow = min(q0 + 32 * u + r, N - 1);
        const op16 *qp = qkv + ((size_t)b * N + qrow) * rs + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[u][s] = *reinterpret_cast<const op16x8 *>(qp + 16 * s);
    }
    op16x8 a_aug, qaug[2];
#pragma unroll
    for (int j = 0; j < 8; j++) { a_aug[j] = (op16)0.f; qaug[0][j] = (op16)0.f; qaug[1][j] = (op16)0.f; }
    if (AUGM) {
        if (hh == 0) a_aug[0] = (op16)1.f;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) qf[u][s][j] = (op16)((float)qf[u][s][j] * scale_log2e);
    }
    f32x16 o[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int i = 0; i < 16; i++) o[u][d][i] = 0.f;
    float m[2] = {AUGM ? 0.f : -INFINITY, AUGM ? 0.f : -INFINITY}, l[2] = {0.f, 0.f};
    const op16 *kbase = qkv + (size_t)b * N * rs + (size_t)H * HD + h * HD;
    const op16 *vbase = kbase + (size_t)H * HD;
    const int swz = (r >> 1) & 7;
    // the 64 NW threads stage 512 + 512 16-byte chunks per tile (8 / NW + 8 / NW per thread).  No register prefetch here: several of these
    // workgroups share a CU, so another workgroup computes while this one waits for its tile.
    const int srow = tid >> 3, sslot = tid & 7;          // chunk c = tid + 64 NW i -> row srow + 8 NW i, slot sslot
    constexpr int PW = 8 / NW;                            // 1-KiB pieces of the K tile (and of the V tile) per wave
    // DMA staging: wave w fills the 1-KiB pieces PW w .. PW w + PW - 1 (8 rows each) of the K tile and of the V tile
    __amdgpu_buffer_rsrc_t rsK, rsV;
    unsigned voK[4], voV[4];                             // (PW of them used; a dependent array size here loses the host stub of the kernel: hipcc 7.2)
    if (DMA) {
        const int nbytes = (int)(((size_t)(N - 1) * rs + HD) * 2);                 // last valid byte of this (batch, head)'s K / V rows
        rsK = __builtin_amdgcn_make_buffer_rsrc((void *)kbase, 0, nbytes, 0x00020000);
        rsV = __builtin_amdgcn_make_buffer_rsrc((void *)vbase, 0, nbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < PW; i++) {
            const int row = 8 * (PW * wave + i) + (lane >> 3), ps = lane & 7;
            voK[i] = (unsigned)((row * rs + (ps ^ ((row >> 1) & 7)) * 8) * 2);
            voV[i] = (unsigned)((row * rs + (ps ^ vsw(row)) * 8) * 2);
        }
    }
    auto dma_tile = [&](int k0, int buf) {
        const unsigned ko = (unsigned)((size_t)k0 * rs * 2);
        unsigned char *kd = smem + buf * 2 * BK * 128 + (PW * wave) * 1024;
#pragma unroll
        for (int i = 0; i < PW; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_void_a *)(kd + i * 1024), 16, voK[i] + ko, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_void_a *)(kd + BK * 128 + i * 1024), 16, voV[i] + ko, 0, 0, 0);
        }
    };
    int ring = 0;
    if (DMA) {
        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    auto tile = [&](int k0, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        if (DMA) {
            Ks = smem + ring * 2 * BK * 128;
            Vs = Ks + BK * 128;
            if (k0 + BK < N) dma_tile(k0 + BK, ring ^ 1);
        }
        if (!DMA) __syncthreads();
#pragma unroll
        for (int i = 0; i < (DMA ? 0 : PW); i++) {
            const int row = srow + 8 * NW * i;
            const uint4 kv = *reinterpret_cast<const uint4 *>(kbase + (size_t)min(k0 + row, N - 1) * rs + sslot * 8);
            const uint4 vv = *reinterpret_cast<const uint4 *>(vbase + (size_t)min(k0 + row, N - 1) * rs + sslot * 8);
            *reinterpret_cast<uint4 *>(Ks + row * 128 + ((sslot ^ ((row >> 1) & 7)) << 4)) = kv;
            *reinterpret_cast<uint4 *>(Vs + row * 128 + ((sslot ^ vsw(row)) << 4)) = vv;
        }
        if (!DMA) __syncthreads();

        if (busy) {          // (a wave whose 64 queries all lie past N only stages its share of the K / V tiles and keeps the barriers)
        op16x2 pk[2][2][8];
        f32x16 sc[2][2];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int i = 0; i < 16; i++) sc[u][kb][i] = 0.f;
        if (AUGM) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                sc[u][0] = COSA_MFMA_32x32x16(a_aug, qaug[u], sc[u][0], 0, 0, 0);
                sc[u][1] = COSA_MFMA_32x32x16(a_aug, qaug[u], sc[u][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int slot = ((2 * s + hh) ^ swz) << 4;
            const op16x8 a0 = *reinterpret_cast<const op16x8 *>(Ks + r * 128 + slot);
            const op16x8 a1 = *reinterpret_cast<const op16x8 *>(Ks + (r + 32) * 128 + slot);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                sc[u][0] = COSA_MFMA_32x32x16(a0, qf[u][s], sc[u][0], 0, 0, 0);
                sc[u][1] = COSA_MFMA_32x32x16(a1, qf[u][s], sc[u][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if constexpr (tail) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (k0 + crow(i, hh) >= N) sc[u][0][i] = -INFINITY;
                    if (k0 + 32 + crow(i, hh) >= N) sc[u][1][i] = -INFINITY;
                }
            }
            // (four independent chains: a single chain of 16 dependent v_max3 stalls on its own latency with two waves per SIMD)
            float mc[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                mc[c] = max3f(sc[u][0][4 * c], sc[u][1][4 * c], sc[u][0][4 * c + 1]);
                mc[c] = max3f(mc[c], sc[u][1][4 * c + 1], sc[u][0][4 * c + 2]);
                mc[c] = max3f(mc[c], sc[u][1][4 * c + 2], sc[u][0][4 * c + 3]);
            }
            float mt = max3f(max3f(mc[0], sc[u][1][3], sc[u][1][7]), max3f(mc[1], sc[u][1][11], sc[u][1][15]), max3f(mc[2], mc[3], mc[0]));
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            if constexpr (AUGM) {
                // the accumulators are already  s * scale * log2 e - m[u]; the reference moves (rarely: deferred as below) by a correction of
                // the accumulators, the running sums and the Q-side constant of the 65th index.  First tile: m = 0 is not a maximum yet.
                const bool first = k0 == 0;
                if (first || __any(mt > kDeferLog2)) {
                    // The new reference must be an op16 value (it enters the score MFMAs as an operand) and must not round back BELOW its
                    // target: with 8-bit operands and |m| >= 1024 the spacing (8) exceeds the deferral threshold, m + mt would round to m,
                    // delta = 0, and the probabilities would keep growing past 2^kDeferLog2.  Adding half a unit of the
                    // operand type's coarsest relative spacing before the rounding makes it round to >= the target, so every move
                    // brings the row maxima back to <= 0.
                    const float tgt = m[u] + fmaxf(mt, 0.f);
                    const float mq = (float)(op16)(first ? mt : tgt + __builtin_fabsf(tgt) * kOp16HalfSpacing);
                    const float delta = mq - m[u];
                    if (!first) {
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        l[u] *= alpha;
#pragma unroll
                        for (int i = 0; i < 16; i++) { o[u][0][i] *= alpha; o[u][1][i] *= alpha; }
                    }
                    m[u] = mq;
                    if (hh == 0) qaug[u][0] = (op16)(-mq);
#pragma unroll
                    for (int i = 0; i < 16; i++) { sc[u][0][i] -= delta; sc[u][1][i] -= delta; }
                }
            } else {
                const float mnew = fmaxf(m[u], mt * scale_log2e);
                // deferred rescale: the running reference m only moves when some row's max has outgrown it by more than 2^DEFER (probabilities
                // then reach at most 2^DEFER, far inside fp32 / op16 range; l and O carry the same factor, so the result is unchanged)
                if (__any(mnew > m[u] + kDeferLog2)) {
                    const float alpha = __builtin_amdgcn_exp2f(m[u] - mnew);
                    l[u] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; i++) { o[u][0][i] *= alpha; o[u][1][i] *= alpha; }
                    m[u] = mnew;
                }
            }
            // probabilities: exp2, rounded to the operand type in pairs (they are the B operand of the PV MFMAs as they stand).  Row sum: fp32
            // over the unrounded values in the order rounds 1-3 fixed (the passes that keep LSE for a backward: the student's results do not
            // move when this kernel is touched), or -- AUGM, no-grad passes -- over the ROUNDED values, what the PV product weights V with, two
            // per instruction
            if constexpr (AUGM) {
                float lp[4] = {0.f, 0.f, 0.f, 0.f};          // (four partial sums: independent v_dot2c chains)
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        pk[u][kb][t] = (op16x2){(op16)__builtin_amdgcn_exp2f(sc[u][kb][2 * t]), (op16)__builtin_amdgcn_exp2f(sc[u][kb][2 * t + 1])};
                        lp[t & 3] = pair_sum(pk[u][kb][t], lp[t & 3]);
                    }
                l[u] += (lp[0] + lp[1]) + (lp[2] + lp[3]);
            } else {
                float ls = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    sc[u][0][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][0][i], scale_log2e, -m[u]));
                    sc[u][1][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][1][i], scale_log2e, -m[u]));
                    ls += sc[u][0][i] + sc[u][1][i];
                }
                l[u] += ls;
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int t = 0; t < 8; t++) pk[u][kb][t] = (op16x2){(op16)sc[u][kb][2 * t], (op16)sc[u][kb][2 * t + 1]};
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int sp = 0; sp < 2; sp++) {
                const int keyb = kb * 32 + 16 * sp + 4 * hh;
                const op16x8 v0 = v_frag(Vs, keyb, 0, lane), v1 = v_frag(Vs, keyb, 1, lane);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const op16x8 pf = __builtin_shufflevector(__builtin_shufflevector(pk[u][kb][4 * sp], pk[u][kb][4 * sp + 1], 0, 1, 2, 3),
                                                              __builtin_shufflevector(pk[u][kb][4 * sp + 2], pk[u][kb][4 * sp + 3], 0, 1, 2, 3),
                                                              0, 1, 2, 3, 4, 5, 6, 7);
                    o[u][0] = COSA_MFMA_32x32x16(v0, pf, o[u][0], 0, 0, 0);
                    o[u][1] = COSA_MFMA_32x32x16(v1, pf, o[u][1], 0, 0, 0);
                }
            }
        }
        if (DMA) {                                     // the next tile has landed and nobody still reads this one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ring ^= 1;
        }
    };
    const int nfull = (N / BK) * BK;
    for (int k0 = 0; k0 < nfull; k0 += BK) tile(k0, std::false_type{});
    if (nfull < N) tile(nfull, std::true_type{});
#pragma unroll
    for (int u = 0; u < 2; u++) {
        float lt = l[u] + __shfl_xor(l[u], 32, 64);
        const float inv = 1.0f / lt;
        const int q = q0 + 32 * u + r;
        if (q < N && C4OUT) {
#if COSA_OP_F16
            const int D = H * HD;
            unsigned char *row = reinterpret_cast<unsigned char *>(out) + ((size_t)b * N + q) * (size_t)(4 * D + 128);
            const int rabs = row0 + b * N + q, Kq = D >> 7;
#pragma unroll
            for (int gp = 0; gp < 2; gp++)
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    unsigned hiw[2][2];
                    float hv[2][4], lv[2][4], amax = 0.f;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int g = 2 * gp + e;
                        _Float16 hi[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            float v = o[u][d][4 * g + j] * inv;
                            asm volatile("" : "+v"(v));          // one rounded fp32 value before the split (see attn_fwd_x3_kernel's output stage)
                            hi[j] = (_Float16)v;
                            hv[e][j] = (float)hi[j];
                            lv[e][j] = (v - hv[e][j]) * kC4LoScale;
                            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(hv[e][j]), __builtin_fabsf(lv[e][j])));
                        }
                        hiw[e][0] = __builtin_bit_cast(unsigned, (op16x2){hi[0], hi[1]});
                        hiw[e][1] = __builtin_bit_cast(unsigned, (op16x2){hi[2], hi[3]});
                    }
                    const unsigned au = __builtin_bit_cast(unsigned, amax);
                    const auto am = __builtin_amdgcn_permlane32_swap(au, au, false, false);          // (non-negative floats order like their bits)
                    const int ex = c4_block_exp(__builtin_bit_cast(float, am[0] > am[1] ? am[0] : am[1]));
                    const float sc = c4_pow2(ex);
                    unsigned LO = 0, HI = 0;                 // [chunk e = 0 (16 bits) | chunk e = 1 (16 bits)]
                    LO = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(LO, lv[0][0], lv[0][1], sc, 0);
                    LO = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(LO, lv[0][2], lv[0][3], sc, 1);
                    LO = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(LO, lv[1][0], lv[1][1], sc, 2);
                    LO = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(LO, lv[1][2], lv[1][3], sc, 3);
                    HI = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(HI, hv[0][0], hv[0][1], sc, 0);
                    HI = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(HI, hv[0][2], hv[0][3], sc, 1);
                    HI = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(HI, hv[1][0], hv[1][1], sc, 2);
                    HI = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(HI, hv[1][2], hv[1][3], sc, 3);
                    const auto a0 = __builtin_amdgcn_permlane32_swap(hiw[0][0], hiw[1][0], false, false);
                    const auto a1 = __builtin_amdgcn_permlane32_swap(hiw[0][1], hiw[1][1], false, false);
                    // hh = 0: P = own LO, Q = the partner's LO;  hh = 1: P = the partner's HI, Q = own HI  (columns 0-3 | 4-7 | 8-11 | 12-15 of the block
                    // are chunk e = 0 of hh = 0, e = 0 of hh = 1, e = 1 of hh = 0, e = 1 of hh = 1)
                    const auto pq = __builtin_amdgcn_permlane32_swap(LO, HI, false, false);
                    const unsigned w0 = (pq[0] & 0xffffu) | (pq[1] << 16), w1 = (pq[0] >> 16) | (pq[1] & 0xffff0000u);
                    const int col = h * HD + 32 * d + 16 * gp;
                    *reinterpret_cast<uint4 *>(row + 2 * (col + 8 * hh)) = make_uint4(a0[0], a1[0], a0[1], a1[1]);
                    *reinterpret_cast<uint2 *>(row + 2 * D + col + 8 * hh) = make_uint2(w0, w1);
                    if (hh == 0) out_scales[c4_scale_off_x(rabs, col >> 7, (col & 127) >> 4, Kq)] = (unsigned char)c4_scale_byte(ex, 0);
                }
            if (h == 0) {                       // augmentation block (1, 1, 0, ...)
                uint4 z = {0u, 0u, 0u, 0u}, one = {0x3c003c00u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < 4; c++) *reinterpret_cast<uint4 *>(row + 4 * D + 64 * hh + 16 * c) = (hh == 0 && c == 0) ? one : z;
            }
            if (hh == 0 && lse) lse[((size_t)b * H + h) * N + q] = (m[u] + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
#endif
        } else if (q < N && C8OUT) {
#if COSA_OP_F16
            const int D = H * HD;
            unsigned char *row = reinterpret_cast<unsigned char *>(out) + ((size_t)b * N + q) * (size_t)(4 * D + 128);
            // a lane holds 4-column chunks (8g + 4hh); v_permlane32_swap with its partner lane (same query, other half) turns two of them into
            // 8 consecutive columns: lanes 0-31 take chunk pair g = 2gp, lanes 32-63 g = 2gp + 1 -- 16-byte fp16 stores, 8-byte stores of the planes
#pragma unroll
            for (int gp = 0; gp < 2; gp++)
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    unsigned hiw[2][2], lo8[2], hi8[2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int g = 2 * gp + e;
                        const float v[4] = {o[u][d][4 * g] * inv, o[u][d][4 * g + 1] * inv, o[u][d][4 * g + 2] * inv, o[u][d][4 * g + 3] * inv};
                        _Float16 hi[4];
                        c8_split4(v, hi, lo8[e], hi8[e]);
                        hiw[e][0] = __builtin_bit_cast(unsigned, (op16x2){hi[0], hi[1]});
                        hiw[e][1] = __builtin_bit_cast(unsigned, (op16x2){hi[2], hi[3]});
                    }
                    const auto a0 = __builtin_amdgcn_permlane32_swap(hiw[0][0], hiw[1][0], false, false);
                    const auto a1 = __builtin_amdgcn_permlane32_swap(hiw[0][1], hiw[1][1], false, false);
                    const auto bl = __builtin_amdgcn_permlane32_swap(lo8[0], lo8[1], false, false);
                    const auto bh = __builtin_amdgcn_permlane32_swap(hi8[0], hi8[1], false, false);
                    const int col = h * HD + 32 * d + 8 * (2 * gp + hh);
                    *reinterpret_cast<uint4 *>(row + 2 * col) = make_uint4(a0[0], a1[0], a0[1], a1[1]);
                    *reinterpret_cast<uint2 *>(row + 2 * D + col) = make_uint2(bl[0], bl[1]);
                    *reinterpret_cast<uint2 *>(row + 3 * D + col) = make_uint2(bh[0], bh[1]);
                }
            if (h == 0) {                       // augmentation block (1, 1, 0, ...): 64 fp16, a half per lane half
                uint4 z = {0u, 0u, 0u, 0u}, one = {0x3c003c00u, 0u, 0u, 0u};
#pragma unroll
                for (int c = 0; c < 4; c++) *reinterpret_cast<uint4 *>(row + 4 * D + 64 * hh + 16 * c) = (hh == 0 && c == 0) ? one : z;
            }
            if (hh == 0 && lse) lse[((size_t)b * H + h) * N + q] = (m[u] + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
#endif
        } else if (q < N) {
            op16 *op = out + ((size_t)b * N + q) * H * HD + h * HD;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                op16x4 v0, v1;
#pragma unroll
                for (int j = 0; j < 4; j++) { v0[j] = (op16)(o[u][0][4 * g + j] * inv); v1[j] = (op16)(o[u][1][4 * g + j] * inv); }
                *reinterpret_cast<op16x4 *>(op + 8 * g + 4 * hh) = v0;
                *reinterpret_cast<op16x4 *>(op + 32 + 8 * g + 4 * hh) = v1;
            }
            if (hh == 0) lse[((size_t)b * H + h) * N + q] = (m[u] + __builtin_amdgcn_logf(lt)) * 0.6931471805599453f;
        }
    }
    if (stamps) {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&stamps[2 * (blockIdx.x & 63) + 1], __builtin_amdgcn_s_memrealtime());
    }
}

// =====================================================================================================
// backward (student pass).  Two MFMA kernels, no atomics, deterministic:
//   attn_bwd_dq_kernel   workgroup owns 128 queries (query on the lane), walks the keys:
//        S^T = K Q^T,  P^T = exp(S^T - lse),  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta) * scale,
//        dQ^T += K^T dS^T
//   attn_bwd_dkv_kernel  workgroup owns 128 keys (key on the lane), walks the queries:
//        S = Q K^T,  P,  dP = dO V^T,  dS,   dV^T += dO^T P,   dK^T += Q^T dS
// In both, the exponentiated / gradient score tile is consumed straight from the accumulator registers
// as the B operand of the next MFMA (rows of the tile are the contraction index).  The operands that
// must be contraction-contiguous along tokens (K^T, Q^T, dO^T) come from a transposing prep kernel,
// which also produces delta = rowsum(dO o O).
// =====================================================================================================

// One LDS image serves both kinds of read of a [64 tokens][64 d] tile (128-B rows, as the rows lie in qkv / dO): 16-B chunk c of
// row r sits at  c ^ usw(r),  usw(r) = ((r>>1)&1)<<2 | (r>>2)&3.  (a) ds_read_b128 row fragments (8 consecutive d of a token: the
// operands of S and dP) are conflict-free: the 16 rows of a lane group give 16 distinct (r&1, usw(r)) pairs; (b) the transposing
// read ds_read_b64_tr_b16 (4 tokens x 16 d per 16-lane group: the K^T / Q^T / dO^T operands) is conflict-free: rows r, r+1 are
// 32 banks apart and bit 2 of usw moves rows r+2, r+3 by 16 banks.  No transposed copies in HBM, no transposed LDS images.
__device__ __forceinline__ int usw(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

__device__ __forceinline__ op16x8 ufrag_rows(const unsigned char *tile, int row, int s, int hh)
{
    return *reinterpret_cast<const op16x8 *>(tile + row * 128 + (((2 * s + hh) ^ usw(row)) << 4));
}

// A operand of a 32x32x16 MFMA whose rows are d (dhalf*32 + lane&31) and whose 8 k-slots are tokens tokb+{0..3, 8..11}
__device__ __forceinline__ op16x8 ufrag_cols(const unsigned char *tile, int tokb, int dhalf, int lane)
{
    const int nn = lane & 15, grp = (lane >> 4) & 1;
    const int r0 = tokb + (nn >> 2), r1 = r0 + 8;
    const int ch = dhalf * 4 + grp * 2 + ((nn & 3) >> 1), off = 8 * (nn & 1);
    union { s16x4v h[2]; op16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4v *)(tile + r0 * 128 + ((ch ^ usw(r0)) << 4) + off));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4v *)(tile + r1 * 128 + ((ch ^ usw(r1)) << 4) + off));
    return u.v;
}

// LDS-DMA of one [64 rows][64 d] tile (8 one-KiB pieces; wave w of 4 issues pieces 2w, 2w+1).  vo[i]: per-lane source offset of
// piece 2w+i with the swizzle applied on the source side; rows past the last valid one read as zeros (buffer range check).
__device__ __forceinline__ void tile_offsets(unsigned vo[2], size_t row_stride, int wave, int lane)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (2 * wave + i) + (lane >> 3), ps = lane & 7;
        vo[i] = (unsigned)((row * row_stride + (size_t)((ps ^ usw(row)) * 8)) * 2);
    }
}
__device__ __forceinline__ void tile_dma(__amdgpu_buffer_rsrc_t rs, unsigned char *dst, const unsigned vo[2], unsigned row0_bytes, int wave)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_a *)(dst + (2 * wave) * 1024), 16, vo[0] + row0_bytes, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_a *)(dst + (2 * wave + 1) * 1024), 16, vo[1] + row0_bytes, 0, 0, 0);
}

// prep: delta[q] = sum_d dO[q,d] * O[q,d]  (the only thing the two kernels below need prepared)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const op16 *__restrict__ dO, const op16 *__restrict__ O,
                                                           float *__restrict__ delta, int N, int H, size_t total)
{
    // one thread per (token, head) quarter: 4 lanes x 16 d
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t th = id >> 2;                                   // (b*N + tok)*H + h
    const int part = (int)(id & 3);
    float acc = 0.f;
    if (th < total) {
        const size_t o = th * HD + part * 16;
        const op16x8 d0 = *reinterpret_cast<const op16x8 *>(dO + o), d1 = *reinterpret_cast<const op16x8 *>(dO + o + 8);
        const op16x8 o0 = *reinterpret_cast<const op16x8 *>(O + o), o1 = *reinterpret_cast<const op16x8 *>(O + o + 8);
#pragma unroll
        for (int i = 0; i < 8; i++) acc += (float)d0[i] * (float)o0[i] + (float)d1[i] * (float)o1[i];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && th < total) {
        const size_t bt = th / H;
        const int h = (int)(th - bt * H);
        const size_t b = bt / N;
        const int tok = (int)(bt - b * N);
        delta[(b * H + h) * N + tok] = acc;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const op16 *__restrict__ qkv, const op16 *__restrict__ dO,
                                                         const float *__restrict__ lse, const float *__restrict__ delta,
                                                         op16 *__restrict__ dqkv, int N, int H, int nblk, int ngroups, float scale)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BK * 128];        // 2-deep ring of (K tile | V tile)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    int blk, b, h;
    if (!attn_block_map(nblk, ngroups, H, blk, b, h)) return;
    const int q0 = blk * BQ + wave * 32;
    const size_t rs = (size_t)3 * H * HD;
    const int qrow = min(q0 + r, N - 1);
    const op16 *qp = qkv + ((size_t)b * N + qrow) * rs + h * HD + 8 * hh;
    const op16 *dop = dO + ((size_t)b * N + qrow) * H * HD + h * HD + 8 * hh;
    op16x8 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; s++) { qf[s] = *reinterpret_cast<const op16x8 *>(qp + 16 * s); dof[s] = *reinterpret_cast<const op16x8 *>(dop + 16 * s); }
    const float scale_log2e = scale * 1.4426950408889634f;
    const float lse2 = lse[((size_t)b * H + h) * N + qrow] * 1.4426950408889634f;
    const float dl = delta[((size_t)b * H + h) * N + qrow];
    f32x16 g0, g1;
#pragma unroll
    for (int i = 0; i < 16; i++) { g0[i] = 0.f; g1[i] = 0.f; }
    const op16 *kbase = qkv + (size_t)b * N * rs + (size_t)H * HD + h * HD;
    const int nbytes = (int)(((size_t)(N - 1) * rs + HD) * 2);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void *)kbase, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void *)(kbase + (size_t)H * HD), 0, nbytes, 0x00020000);
    unsigned vo[2];
    tile_offsets(vo, rs, wave, lane);
    auto dma = [&](int k0, int buf) {
        const unsigned ro = (unsigned)((size_t)k0 * rs * 2);
        tile_dma(rsK, smem + buf * 2 * BK * 128, vo, ro, wave);
        tile_dma(rsV, smem + buf * 2 * BK * 128 + BK * 128, vo, ro, wave);
    };
    int ring = 0;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](int k0, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        const unsigned char *Ks = smem + ring * 2 * BK * 128, *Vsr = Ks + BK * 128;
        if (k0 + BK < N) dma(k0 + BK, ring ^ 1);
        f32x16 s0, s1, p0, p1;
#pragma unroll
        for (int i = 0; i < 16; i++) { s0[i] = 0.f; s1[i] = 0.f; p0[i] = 0.f; p1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            s0 = COSA_MFMA_32x32x16(ufrag_rows(Ks, r, s, hh), qf[s], s0, 0, 0, 0);
            s1 = COSA_MFMA_32x32x16(ufrag_rows(Ks, r + 32, s, hh), qf[s], s1, 0, 0, 0);
            p0 = COSA_MFMA_32x32x16(ufrag_rows(Vsr, r, s, hh), dof[s], p0, 0, 0, 0);
            p1 = COSA_MFMA_32x32x16(ufrag_rows(Vsr, r + 32, s, hh), dof[s], p1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            float a = __builtin_amdgcn_exp2f(s0[i] * scale_log2e - lse2);
            float c = __builtin_amdgcn_exp2f(s1[i] * scale_log2e - lse2);
            if constexpr (tail) {
                if (k0 + crow(i, hh) >= N) a = 0.f;
                if (k0 + 32 + crow(i, hh) >= N) c = 0.f;
            }
            s0[i] = a * (p0[i] - dl) * scale;        // dS^T
            s1[i] = c * (p1[i] - dl) * scale;
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int sp = 0; sp < 2; sp++) {
                op16x8 df;
#pragma unroll
                for (int j = 0; j < 8; j++) df[j] = (op16)(kb == 0 ? s0[8 * sp + j] : s1[8 * sp + j]);
                const int keyb = kb * 32 + 16 * sp + 4 * hh;
                g0 = COSA_MFMA_32x32x16(ufrag_cols(Ks, keyb, 0, lane), df, g0, 0, 0, 0);     // K^T by transposing reads
                g1 = COSA_MFMA_32x32x16(ufrag_cols(Ks, keyb, 1, lane), df, g1, 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next tile has landed and nobody still reads this one
        __syncthreads();
        ring ^= 1;
    };
    const int nfull = (N / BK) * BK;
    for (int k0 = 0; k0 < nfull; k0 += BK) tile(k0, std::false_type{});
    if (nfull < N) tile(nfull, std::true_type{});
    const int q = q0 + r;
    if (q < N) {
        op16 *op = dqkv + ((size_t)b * N + q) * rs + h * HD;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            op16x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; j++) { v0[j] = (op16)g0[4 * g + j]; v1[j] = (op16)g1[4 * g + j]; }
            *reinterpret_cast<op16x4 *>(op + 8 * g + 4 * hh) = v0;
            *reinterpret_cast<op16x4 *>(op + 32 + 8 * g + 4 * hh) = v1;
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const op16 *__restrict__ qkv, const op16 *__restrict__ dO,
                                                          const float *__restrict__ lse, const float *__restrict__ delta,
                                                          op16 *__restrict__ dqkv, int N, int H, int nblk, int ngroups, float scale)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BK * 128 + 4 * BK * 4];  // ring of (Q | dO) + (lse | delta) x 2
    float *sc_s = reinterpret_cast<float *>(smem + 4 * BK * 128);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    int blk, b, h;
    if (!attn_block_map(nblk, ngroups, H, blk, b, h)) return;
    const int key0 = blk * BQ + wave * 32;
    const size_t rs = (size_t)3 * H * HD;
    const int krow = min(key0 + r, N - 1);
    const op16 *kp = qkv + ((size_t)b * N + krow) * rs + (size_t)H * HD + h * HD + 8 * hh;
    const op16 *vp = qkv + ((size_t)b * N + krow) * rs + (size_t)2 * H * HD + h * HD + 8 * hh;
    op16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; s++) { kf[s] = *reinterpret_cast<const op16x8 *>(kp + 16 * s); vf[s] = *reinterpret_cast<const op16x8 *>(vp + 16 * s); }
    const float scale_log2e = scale * 1.4426950408889634f;
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int i = 0; i < 16; i++) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }
    const op16 *qbase = qkv + (size_t)b * N * rs + h * HD;
    const op16 *dobase = dO + (size_t)b * N * H * HD + h * HD;
    const size_t ds = (size_t)H * HD;
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)qbase, 0, (int)(((size_t)(N - 1) * rs + HD) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)dobase, 0, (int)(((size_t)(N - 1) * ds + HD) * 2), 0x00020000);
    unsigned voQ[2], voD[2];
    tile_offsets(voQ, rs, wave, lane);
    tile_offsets(voD, ds, wave, lane);
    const float *lseb = lse + ((size_t)b * H + h) * N;
    const float *dlb = delta + ((size_t)b * H + h) * N;
    auto dma = [&](int q0, int buf) {
        tile_dma(rsQ, smem + buf * 2 * BK * 128, voQ, (unsigned)((size_t)q0 * rs * 2), wave);
        tile_dma(rsD, smem + buf * 2 * BK * 128 + BK * 128, voD, (unsigned)((size_t)q0 * ds * 2), wave);
        if (tid < 2 * BK) {                                           // log-sum-exp and delta of the tile's 64 queries
            const int i = tid & (BK - 1), q = q0 + i;
            const float v = q < N ? (tid < BK ? lseb[q] * 1.4426950408889634f : dlb[q]) : 0.f;
            sc_s[buf * 2 * BK + tid] = v;
        }
    };
    int ring = 0;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tile = [&](int q0, auto tail_tag) {
        constexpr bool tail = decltype(tail_tag)::value;
        const unsigned char *Qs = smem + ring * 2 * BK * 128, *dOs = Qs + BK * 128;
        const float *lse_s = sc_s + ring * 2 * BK, *dl_s = lse_s + BK;
        if (q0 + BK < N) dma(q0 + BK, ring ^ 1);
        // one 32-query block at a time (keeps S/dP to 32 registers so that two waves fit a SIMD):
        // S[q][key], dP[q][key] (rows = queries in registers, key on the lane), then the dV^T / dK^T updates
#pragma unroll 1
        for (int qb = 0; qb < 2; qb++) {
            f32x16 s0, p0;
#pragma unroll
            for (int i = 0; i < 16; i++) { s0[i] = 0.f; p0[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                s0 = COSA_MFMA_32x32x16(ufrag_rows(Qs, r + 32 * qb, s, hh), kf[s], s0, 0, 0, 0);
                p0 = COSA_MFMA_32x32x16(ufrag_rows(dOs, r + 32 * qb, s, hh), vf[s], p0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int qa = 32 * qb + crow(i, hh);
                float a = __builtin_amdgcn_exp2f(s0[i] * scale_log2e - lse_s[qa]);
                if constexpr (tail) { if (q0 + qa >= N) a = 0.f; }
                s0[i] = a;                                 // P
                p0[i] = a * (p0[i] - dl_s[qa]) * scale;    // dS
            }
#pragma unroll
            for (int sp = 0; sp < 2; sp++) {
                op16x8 pf, df;
#pragma unroll
                for (int j = 0; j < 8; j++) { pf[j] = (op16)s0[8 * sp + j]; df[j] = (op16)p0[8 * sp + j]; }
                const int tokb = qb * 32 + 16 * sp + 4 * hh;
                dv0 = COSA_MFMA_32x32x16(ufrag_cols(dOs, tokb, 0, lane), pf, dv0, 0, 0, 0);   // dO^T, Q^T by transposing reads
                dv1 = COSA_MFMA_32x32x16(ufrag_cols(dOs, tokb, 1, lane), pf, dv1, 0, 0, 0);
                dk0 = COSA_MFMA_32x32x16(ufrag_cols(Qs, tokb, 0, lane), df, dk0, 0, 0, 0);
                dk1 = COSA_MFMA_32x32x16(ufrag_cols(Qs, tokb, 1, lane), df, dk1, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ring ^= 1;
    };
    const int nfull = (N / BK) * BK;
    for (int q0 = 0; q0 < nfull; q0 += BK) tile(q0, std::false_type{});
    if (nfull < N) tile(nfull, std::true_type{});
    const int key = key0 + r;
    if (key < N) {
        op16 *okp = dqkv + ((size_t)b * N + key) * rs + (size_t)H * HD + h * HD;
        op16 *ovp = dqkv + ((size_t)b * N + key) * rs + (size_t)2 * H * HD + h * HD;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            op16x4 a0, a1, c0, c1;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a0[j] = (op16)dk0[4 * g + j]; a1[j] = (op16)dk1[4 * g + j];
                c0[j] = (op16)dv0[4 * g + j]; c1[j] = (op16)dv1[4 * g + j];
            }
            *reinterpret_cast<op16x4 *>(okp + 8 * g + 4 * hh) = a0;
            *reinterpret_cast<op16x4 *>(okp + 32 + 8 * g + 4 * hh) = a1;
            *reinterpret_cast<op16x4 *>(ovp + 8 * g + 4 * hh) = c0;
            *reinterpret_cast<op16x4 *>(ovp + 32 + 8 * g + 4 * hh) = c1;
        }
    }
}

}  // namespace
}  // namespace cosa

using namespace cosa;

#if COSA_OP_F16        // second build of this file (fp16 operands): the same entry points under their _f16 names (include/cosa_hip.h)
#define cosa_attn_workspace_bytes cosa_attn_workspace_bytes_f16
#define cosa_attn_prepare_vt cosa_attn_prepare_vt_f16
#define cosa_attn_fwd cosa_attn_fwd_f16
#define cosa_attn_fwd_bf16x3 cosa_attn_fwd_f16x3
#define cosa_attn_bwd_workspace_bytes cosa_attn_bwd_workspace_bytes_f16
#define cosa_attn_bwd cosa_attn_bwd_f16
#endif

/* scratch of cosa_attn_fwd: none any more (the V^T copy is gone); a token size keeps the calling convention */
extern "C" size_t cosa_attn_workspace_bytes(int B, int N, int H)
{
    (void)B; (void)N; (void)H;
    return 256;
}

extern "C" int cosa_attn_prepare_vt(const void *qkv, int B, int N, int H, void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(qkv && workspace && B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "cosa_attn_prepare_vt: bad arguments");
    if (workspace_bytes < cosa_attn_workspace_bytes(B, N, H)) {
        set_error("cosa_attn_prepare_vt: workspace too small");
        return COSA_ENOMEM;
    }
    (void)stream;            // nothing to prepare: kept so that callers written against the V^T version keep working
    return COSA_OK;
}

// Waves per workgroup of the forward kernel (2 = 128 queries, 4 = 256 queries per workgroup; eight waves per CU either way): a wave's work
// per query block is the same in both, so the launch with fewer rounds of the chip wins -- 1024 resident 2-wave workgroups against 512
// 4-wave ones -- and on a tie the wider workgroup, whose waves issue half the LDS-DMA instructions per tile (measured, b = 32 x 12 heads:
// N = 1765 380 -> 371 us, N = 785 115 -> 112, b = 16 N = 785 64 -> 58, N = 3601 (b = 8) 400 -> 379; N = 1601, where 256-query blocks pad
// more, 319 vs 333: stays narrow).
static bool attn_wide(int B, int N, int H)
{
    if (N <= 512) return false;
    const long groups = (long)(B * H + 7) / 8 * 8;
    const long r2 = (groups * ((N + 127) / 128) + 1023) / 1024, r4 = (groups * ((N + 255) / 256) + 511) / 512;
    return r4 <= r2;
}

extern "C" int cosa_attn_fwd(const void *qkv, void *out, float *lse, int B, int N, int H, int head_dim, float scale,
                             int flags, uint64_t *stamps, void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(qkv && out && lse && workspace, "cosa_attn_fwd: null pointer");
    COSA_REQUIRE(head_dim == HD, "cosa_attn_fwd: head_dim must be 64");
    COSA_REQUIRE(B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "cosa_attn_fwd: bad shape");
    if (workspace_bytes < cosa_attn_workspace_bytes(B, N, H)) {
        set_error("cosa_attn_fwd: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    const int Npad = (N + BK - 1) / BK * BK;
    const op16 *vt = nullptr;      // (there is no V^T copy)
    bool wide = attn_wide(B, N, H);
    if (flags & 0x100) wide = true;          // flags bits 8 / 9 force 4 / 2 waves per workgroup (measurements); the other bits selected among
    if (flags & 0x200) wide = false;         // earlier kernel variants and are ignored
    const int bq = wide ? 256 : 128;
    const int nblk = (N + bq - 1) / bq;
    const dim3 grid(nblk * ((B * H + 7) / 8 * 8));
    const float sl2 = scale * 1.4426950408889634f;
    unsigned long long *stp = reinterpret_cast<unsigned long long *>(stamps);
    const op16 *q = static_cast<const op16 *>(qkv);
    op16 *o = static_cast<op16 *>(out);
    // LDS-DMA addresses one image's qkv rows through a 32-bit buffer offset; beyond 2 GiB per image the tiles are staged through registers
    if ((size_t)N * 3 * H * HD * 2 >= 0x7fffffffull)
        hipLaunchKernelGGL(attn_fwd2_kernel<false>, dim3(((N + 127) / 128) * ((B * H + 7) / 8 * 8)), dim3(128), 0, st, q, vt, o, lse, N, Npad, H, (N + 127) / 128,
                           B * H, sl2, stp);
    else if (wide && (flags & 0x400))          // bit 10: a no-grad pass -- Q may be pre-scaled in the operand type (AUGM above)
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 0, 4, true>), grid, dim3(256), 0, st, q, vt, o, lse, N, Npad, H, nblk, B * H, sl2, stp);
    else if (wide)
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 0, 4>), grid, dim3(256), 0, st, q, vt, o, lse, N, Npad, H, nblk, B * H, sl2, stp);
    else if (flags & 0x400)
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 0, 2, true>), grid, dim3(128), 0, st, q, vt, o, lse, N, Npad, H, nblk, B * H, sl2, stp);
    else
        hipLaunchKernelGGL(attn_fwd2_kernel<true>, grid, dim3(128), 0, st, q, vt, o, lse, N, Npad, H, nblk, B * H, sl2, stp);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

#if COSA_OP_F16
/* attention on plain fp16 qkv rows [B, N, 3, H, 64] whose output leaves as fp16c8 rows [B*N, 4 H 64 + 128 bytes] (c8.hpp) for the c8 output
 * projection; lse optional */
extern "C" int cosa_attn_fwd_f16c8(const void *qkv, void *out_c8, float *lse, int B, int N, int H, int head_dim, float scale,
                                   uint64_t *stamps, void *stream)
{
    COSA_REQUIRE(qkv && out_c8, "cosa_attn_fwd_f16c8: null pointer");
    COSA_REQUIRE(head_dim == HD, "cosa_attn_fwd_f16c8: head_dim must be 64");
    COSA_REQUIRE(B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "cosa_attn_fwd_f16c8: bad shape");
    COSA_REQUIRE((size_t)N * 3 * H * HD * 2 < 0x7fffffffull, "cosa_attn_fwd_f16c8: one image's qkv rows must stay below 2 GiB (buffer addressing)");
    const int Npad = (N + BK - 1) / BK * BK;
    const bool wide = attn_wide(B, N, H);
    const int nblk = wide ? (N + 255) / 256 : (N + 127) / 128;
    if (wide)
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 1, 4, true>), dim3(nblk * ((B * H + 7) / 8 * 8)), dim3(256), 0, as_stream(stream), static_cast<const op16 *>(qkv),
                           static_cast<const op16 *>(nullptr), static_cast<op16 *>(out_c8), lse, N, Npad, H, nblk, B * H, scale * 1.4426950408889634f,
                           reinterpret_cast<unsigned long long *>(stamps));
    else
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 1, 2, true>), dim3(nblk * ((B * H + 7) / 8 * 8)), dim3(128), 0, as_stream(stream), static_cast<const op16 *>(qkv),
                           static_cast<const op16 *>(nullptr), static_cast<op16 *>(out_c8), lse, N, Npad, H, nblk, B * H, scale * 1.4426950408889634f,
                           reinterpret_cast<unsigned long long *>(stamps));
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
#endif

#if COSA_OP_F16
// attention on plain fp16 qkv rows -> fp16c4 rows out_c4 [B*N, 4 H HD + 128 bytes] + their scale bytes in out_scales, the scale tensor of the
// WHOLE operand the rows belong to (cosa_c4_scale_bytes(rows of that operand, H HD)); row0 = index of out_c4's first row in that operand
extern "C" int cosa_attn_fwd_f16c4(const void *qkv, void *out_c4, void *out_scales, int row0, float *lse, int B, int N, int H, int head_dim,
                                   float scale, uint64_t *stamps, void *stream)
{
    COSA_REQUIRE(qkv && out_c4 && out_scales && row0 >= 0, "cosa_attn_fwd_f16c4: null pointer");
    COSA_REQUIRE(head_dim == HD, "cosa_attn_fwd_f16c4: head_dim must be 64");
    COSA_REQUIRE(B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535 && (H * HD) % 256 == 0, "cosa_attn_fwd_f16c4: bad shape (H * 64 %% 256 == 0)");
    COSA_REQUIRE((size_t)N * 3 * H * HD * 2 < 0x7fffffffull, "cosa_attn_fwd_f16c4: one image's qkv rows must stay below 2 GiB (buffer addressing)");
    const int Npad = (N + BK - 1) / BK * BK;
    const bool wide = attn_wide(B, N, H);
    const int nblk = wide ? (N + 255) / 256 : (N + 127) / 128;
    unsigned char *sc = static_cast<unsigned char *>(out_scales);
    if (wide)
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 2, 4, true>), dim3(nblk * ((B * H + 7) / 8 * 8)), dim3(256), 0, as_stream(stream), static_cast<const op16 *>(qkv),
                           static_cast<const op16 *>(nullptr), static_cast<op16 *>(out_c4), lse, N, Npad, H, nblk, B * H, scale * 1.4426950408889634f,
                           reinterpret_cast<unsigned long long *>(stamps), sc, row0);
    else
        hipLaunchKernelGGL((attn_fwd2_kernel<true, 2, 2, true>), dim3(nblk * ((B * H + 7) / 8 * 8)), dim3(128), 0, as_stream(stream), static_cast<const op16 *>(qkv),
                           static_cast<const op16 *>(nullptr), static_cast<op16 *>(out_c4), lse, N, Npad, H, nblk, B * H, scale * 1.4426950408889634f,
                           reinterpret_cast<unsigned long long *>(stamps), sc, row0);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
#endif

/* backward workspace: delta [B,H,N] f32 (the transposed operands come from transposing LDS reads now) */
extern "C" int cosa_attn_fwd_bf16x3(const void *qkv_split, void *out_split, float *lse, int B, int N, int H, int head_dim, float scale,
                                    int ldq, int ldo, uint64_t *stamps, void *stream)
{
    COSA_REQUIRE(qkv_split && out_split && B > 0 && N > 0 && H > 0, "cosa_attn_fwd_bf16x3: bad arguments");
    COSA_REQUIRE(head_dim == HD, "cosa_attn_fwd_bf16x3: head_dim must be 64");
    COSA_REQUIRE(ldq >= 6 * H * HD && ldq % 8 == 0 && ldo >= 2 * H * HD + 64 && ldo % 8 == 0, "cosa_attn_fwd_bf16x3: bad row strides");
    const int nblk = (N + BQ - 1) / BQ, ngroups = B * H;
    const int grid = ((ngroups + 7) / 8) * 8 * nblk;
    COSA_REQUIRE((size_t)N * ldq * 2 < 0x7fffffffull, "cosa_attn_fwd_bf16x3: one image's qkv rows must stay below 2 GB (buffer addressing)");
    constexpr int kLdsX3 = 2 * 4 * BK * 128;
    static bool attr_done = false;
    if (!attr_done) {
        COSA_HIP_CHECK(hipFuncSetAttribute((const void *)attn_fwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsX3));
        attr_done = true;
    }
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3(grid), dim3(256), kLdsX3, as_stream(stream), static_cast<const op16 *>(qkv_split),
                       static_cast<op16 *>(out_split), lse, N, H, nblk, ngroups, scale * 1.4426950408889634f, ldq, ldo,
                       reinterpret_cast<unsigned long long *>(stamps));
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" size_t cosa_attn_bwd_workspace_bytes(int B, int N, int H)
{
    return align_up((size_t)B * H * N * sizeof(float), 256);
}

extern "C" int cosa_attn_bwd(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                             int B, int N, int H, int head_dim, float scale, void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(qkv && out && dout && lse && dqkv && workspace, "cosa_attn_bwd: null pointer");
    COSA_REQUIRE(head_dim == HD, "cosa_attn_bwd: head_dim must be 64");
    COSA_REQUIRE(B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "cosa_attn_bwd: bad shape");
    COSA_REQUIRE((size_t)N * 3 * H * HD * 2 < 0x7fffffffull, "cosa_attn_bwd: one image's qkv rows must stay below 2 GiB");
    if (workspace_bytes < cosa_attn_bwd_workspace_bytes(B, N, H)) {
        set_error("cosa_attn_bwd: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    float *delta = static_cast<float *>(workspace);
    const op16 *q = static_cast<const op16 *>(qkv), *o = static_cast<const op16 *>(out), *d_o = static_cast<const op16 *>(dout);
    const size_t total = (size_t)B * N * H;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, st, d_o, o, delta, N, H, total);
    COSA_LAUNCH_CHECK();
    const int nblk = (N + BQ - 1) / BQ;
    const dim3 grid(nblk * ((B * H + 7) / 8 * 8));
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, q, d_o, lse, delta, static_cast<op16 *>(dqkv), N, H, nblk, B * H, scale);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, q, d_o, lse, delta, static_cast<op16 *>(dqkv), N, H, nblk, B * H, scale);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
