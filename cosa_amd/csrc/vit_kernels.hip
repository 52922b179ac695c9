// vit_kernels.hip -- element-wise pieces of the STUDENT's transformer blocks (training path, bf16 residual stream).
//
//   models/vit/vit.py:154-158   x = x + attn(norm1(x)); x = x + mlp(norm2(x))
//
// cosa_add_layernorm_fwd:  x_new = x + delta (rounded to bf16, the residual stream's dtype), y = LayerNorm(x_new) -- one pass
//                          instead of torch's add kernel + layer_norm kernel; keeps mean / rstd for the backward.
// cosa_layernorm_bwd:      dx = LayerNorm'(dy) + dskip (the gradient that reaches x_new through the skip connection), and the
//                          weight / bias gradients: per-workgroup partial sums, then a small deterministic reduction
//                          (no atomics) -- replaces layer_norm_grad_input + 2 gamma/beta kernels + the gradient add.
// HBM-bound: 8 B per element forward, 8 B backward.  One wave per row, 12 elements per lane (D = 768).
#include "kernels.hpp"
#include "c8.hpp"
#include <cstdlib>

namespace cosa {
namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int D = 768, PER = 3;            // 3 x 4 consecutive elements per lane: columns (lane + 64 i) * 4

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ delta,
                                                        const bf16 *__restrict__ g, const bf16 *__restrict__ b,
                                                        bf16 *__restrict__ xout, bf16 *__restrict__ y, float *__restrict__ mean_o,
                                                        float *__restrict__ rstd_o, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)row * D;
    float v[PER][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const bf16x4 xv = *reinterpret_cast<const bf16x4 *>(x + base + c0);
        bf16x4 sv = xv;
        if (delta) {
            const bf16x4 dv = *reinterpret_cast<const bf16x4 *>(delta + base + c0);
#pragma unroll
            for (int j = 0; j < 4; j++) sv[j] = (bf16)((float)xv[j] + (float)dv[j]);       // the stream is bf16: LN sees the rounded sum
            if (xout) *reinterpret_cast<bf16x4 *>(xout + base + c0) = sv;
        } else if (xout && xout != x) {
            *reinterpret_cast<bf16x4 *>(xout + base + c0) = sv;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { v[i][j] = (float)sv[j]; s += v[i][j]; }
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { const float a = v[i][j] - mean; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const bf16x4 gg = *reinterpret_cast<const bf16x4 *>(g + c0);
        const bf16x4 bb = *reinterpret_cast<const bf16x4 *>(b + c0);
        bf16x4 ov;
#pragma unroll
        for (int j = 0; j < 4; j++) ov[j] = (bf16)((v[i][j] - mean) * rstd * (float)gg[j] + (float)bb[j]);
        *reinterpret_cast<bf16x4 *>(y + base + c0) = ov;
    }
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// part: [gridDim.x][2][D] (dgamma partials, dbeta partials)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16 *__restrict__ dy, const bf16 *__restrict__ xn,
                                                    const float *__restrict__ mean, const float *__restrict__ rstd,
                                                    const bf16 *__restrict__ g, const bf16 *__restrict__ dskip, bf16 *__restrict__ dx,
                                                    float *__restrict__ part, int rows, int rows_per_wg)
{
    __shared__ float red[4][2][D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_wg;
    int r1 = r0 + rows_per_wg;
    r1 = r1 < rows ? r1 : rows;
    float gam[PER][4], dg[PER][4], db[PER][4];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const bf16x4 gg = *reinterpret_cast<const bf16x4 *>(g + (lane + 64 * i) * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { gam[i][j] = (float)gg[j]; dg[i][j] = 0.f; db[i][j] = 0.f; }
    }
    // software-pipelined over the wave's rows: the next row's three loads are in flight while this row is reduced (a wave owns
    // ~12 rows and its reductions are serial, so without the prefetch the kernel is load-latency-bound at ~2 TB/s)
    bf16x4 ndy[PER], nxn[PER], nsk[PER];
    auto load_row = [&](int row) {
        const size_t base = (size_t)row * D;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c0 = (lane + 64 * i) * 4;
            ndy[i] = *reinterpret_cast<const bf16x4 *>(dy + base + c0);
            nxn[i] = *reinterpret_cast<const bf16x4 *>(xn + base + c0);
            if (dskip) nsk[i] = *reinterpret_cast<const bf16x4 *>(dskip + base + c0);
        }
    };
    int row = r0 + wave;
    if (row < r1) load_row(row);
    for (; row < r1; row += 4) {
        const size_t base = (size_t)row * D;
        const float mu = mean[row], rs = rstd[row];
        bf16x4 cdy[PER], cxn[PER], csk[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) { cdy[i] = ndy[i]; cxn[i] = nxn[i]; csk[i] = nsk[i]; }
        if (row + 4 < r1) load_row(row + 4);
        float xh[PER][4], gy[PER][4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float d = (float)cdy[i][j];
                xh[i][j] = ((float)cxn[i][j] - mu) * rs;
                gy[i][j] = d * gam[i][j];
                c1 += gy[i][j];
                c2 += gy[i][j] * xh[i][j];
                dg[i][j] += d * xh[i][j];
                db[i][j] += d;
            }
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c0 = (lane + 64 * i) * 4;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (gy[i][j] - c1 - xh[i][j] * c2) * rs;
            if (dskip) {
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] += (float)csk[i][j];
            }
            bf16x4 ov;
#pragma unroll
            for (int j = 0; j < 4; j++) ov[j] = (bf16)o[j];
            *reinterpret_cast<bf16x4 *>(dx + base + c0) = ov;
        }
    }
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            red[wave][0][(lane + 64 * i) * 4 + j] = dg[i][j];
            red[wave][1][(lane + 64 * i) * 4 + j] = db[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int k = c / D, col = c - k * D;
        part[((size_t)blockIdx.x * 2 + k) * D + col] = red[0][k][col] + red[1][k][col] + red[2][k][col] + red[3][k][col];
    }
}

// ---- fp32 residual stream (round 4): the student's stream is fp32 like the reference's (main.py:124-246 runs without autocast) --------
// Forward of a block's LayerNorm is the teacher's layernorm_kernel (fp32 row in, bf16 row out); the residual add happens in the
// projection GEMM's fp32 epilogue.  Backward: dx (fp32) = LayerNorm'(dy) + dskip (fp32: the gradient that reaches the same stream
// tensor through the skip connection), plus a bf16 copy dx16 of that sum -- the dY operand of the previous projection's input- and
// weight-gradient GEMMs -- so the sum is formed once, in fp32, and never re-read for a cast.  mean / rstd are recomputed from the fp32
// row with the forward kernel's own operation order (the row is in registers anyway), so nothing but the stream itself is saved.
// Traffic: 2 (dy) + 4 (x) + 4 (dskip) in, 4 (dx) + 2 (dx16) out = 16 B per element.
// DY32: dy is fp32 (the final norm of the training path, whose output gradient is the fp32 sum of the decoder's and the heads' gradients)
template <bool DY32>
__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const void *__restrict__ dyv, const float *__restrict__ x,
                                                        const bf16 *__restrict__ g, const float *__restrict__ dskip,
                                                        float *__restrict__ dx, bf16 *__restrict__ dx16, float *__restrict__ part,
                                                        int rows, int rows_per_wg, float eps)
{
    __shared__ float red[4][2][D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_wg;
    int r1 = r0 + rows_per_wg;
    r1 = r1 < rows ? r1 : rows;
    float gam[PER][4], dg[PER][4], db[PER][4];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const bf16x4 gg = *reinterpret_cast<const bf16x4 *>(g + (lane + 64 * i) * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { gam[i][j] = (float)gg[j]; dg[i][j] = 0.f; db[i][j] = 0.f; }
    }
    float4 ndy[PER];
    float4 nx[PER], nsk[PER];
    auto load_row = [&](int row) {
        const size_t base = (size_t)row * D;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c0 = (lane + 64 * i) * 4;
            if (DY32) {
                ndy[i] = *reinterpret_cast<const float4 *>(static_cast<const float *>(dyv) + base + c0);
            } else {
                const bf16x4 t = *reinterpret_cast<const bf16x4 *>(static_cast<const bf16 *>(dyv) + base + c0);
                ndy[i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
            }
            nx[i] = *reinterpret_cast<const float4 *>(x + base + c0);
            if (dskip) nsk[i] = *reinterpret_cast<const float4 *>(dskip + base + c0);
        }
    };
    int row = r0 + wave;
    if (row < r1) load_row(row);
    for (; row < r1; row += 4) {
        const size_t base = (size_t)row * D;
        float4 cdy[PER];
        float4 cx[PER], csk[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) { cdy[i] = ndy[i]; cx[i] = nx[i]; csk[i] = nsk[i]; }
        if (row + 4 < r1) load_row(row + 4);
        // the forward kernel's statistics, operation for operation (layernorm_kernel, gemm_kernels.hip)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; i++) s += cx[i].x + cx[i].y + cx[i].z + cx[i].w;
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const float a = cx[i].x - mu, c = cx[i].y - mu, d = cx[i].z - mu, e = cx[i].w - mu;
            q += a * a + c * c + d * d + e * e;
        }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
        float xh[PER][4], gy[PER][4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const float xv[4] = {cx[i].x, cx[i].y, cx[i].z, cx[i].w};
            const float dv[4] = {cdy[i].x, cdy[i].y, cdy[i].z, cdy[i].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float d = dv[j];
                xh[i][j] = (xv[j] - mu) * rs;
                gy[i][j] = d * gam[i][j];
                c1 += gy[i][j];
                c2 += gy[i][j] * xh[i][j];
                dg[i][j] += d * xh[i][j];
                db[i][j] += d;
            }
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c0 = (lane + 64 * i) * 4;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (gy[i][j] - c1 - xh[i][j] * c2) * rs;
            if (dskip) { o[0] += csk[i].x; o[1] += csk[i].y; o[2] += csk[i].z; o[3] += csk[i].w; }
            *reinterpret_cast<float4 *>(dx + base + c0) = make_float4(o[0], o[1], o[2], o[3]);
            if (dx16) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; j++) ov[j] = (bf16)o[j];
                *reinterpret_cast<bf16x4 *>(dx16 + base + c0) = ov;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            red[wave][0][(lane + 64 * i) * 4 + j] = dg[i][j];
            red[wave][1][(lane + 64 * i) * 4 + j] = db[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int k = c / D, col = c - k * D;
        part[((size_t)blockIdx.x * 2 + k) * D + col] = red[0][k][col] + red[1][k][col] + red[2][k][col] + red[3][k][col];
    }
}

// dgamma / dbeta [D] = sum over the workgroup partials, in a fixed order (deterministic).  grid = 2*D/8, 256 threads:
// 8 columns x 32 slices of the partial list (8 serial loads per thread), slices combined through LDS.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float *__restrict__ part, int nblk, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, int accumulate)
{
    __shared__ float red[32][8];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;                  // c in [0, 2D)
    const int k = c / D, col = c - k * D;
    float s = 0.f;
    for (int p = sl; p < nblk; p += 32) s += part[((size_t)p * 2 + k) * D + col];
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++) t += red[i][cl];
        float *dst = k == 0 ? dgamma : dbeta;
        dst[col] = accumulate ? dst[col] + t : t;
    }
}

// ---- narrow heads: Y[M, N<=32] (fp32) = X[M, K] W[N, K]^T -------------------------------------------------------------------
// The CAM / auxiliary-CAM heads (1x1 conv 768 -> 20|80, models/__init__.py:190-192), LargeFOV's conv8 (512 -> 21|81) and the
// classification heads are "skinny" GEMMs: a library GEMM pads N to its tile and picks tile / split by the row count, so results
// move in their last bits with the batch.  Here a workgroup owns 16 token rows and runs them through the EXACT-fp32 matrix
// instruction v_mfma_f32_16x16x4_f32 (bit for bit an fmaf chain in k order): X is read once, as whole lines -- lane (row r, q)
// loads the 16 consecutive elements X[r][64j + 16q ..] -- and MFMA step i of a 64-wide k block contracts the elements 16q + i of the
// four lane quarters.  The 64-wide k blocks are dealt round-robin to the four waves (wave w: blocks w, w + 4, ...), whose partial
// sums meet in LDS and are added in wave order, so a 16-row group keeps 4 waves' worth of loads in flight and a launch has
// M / 16 workgroups (12 waves per CU at the benchmark's 12.5k rows) -- the first version kept the weights in 98 KB of LDS (one
// workgroup per CU, one wave per row group, a serial fill) and took 93 us on a 5 us problem.  The weights (<= 32 x K, <= 96 KB) are
// read through the same kind of loads (lane (col, q) -> W[col][64j + 16q ..]) and stay L2-resident.  HBM-bound on X; the
// reduction order of a row is fixed, so the result of a token does not depend on the batch around it.  16-bit inputs are widened.
typedef float f32x4h __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void head_load16(const T *p, float (&v)[16])
{
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 f = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + 4 * c);
            v[4 * c] = f.x; v[4 * c + 1] = f.y; v[4 * c + 2] = f.z; v[4 * c + 3] = f.w;
        }
    } else {
        typedef T t8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const t8 f = *reinterpret_cast<const t8 *>(p + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; e++) v[8 * c + e] = (float)f[e];
        }
    }
}

template <typename T, int NB>       // NB: 16-column blocks of the output per slice (1 or 2); N > 16 NB: the slices are a loop inside the kernel
__global__ __launch_bounds__(256, 2) void head_gemm_kernel(const T *__restrict__ X, const T *__restrict__ W, float *__restrict__ Y,
                                                          int M, int N, int K, int rows_per_img, long long img_stride, int ldx,
                                                          int round_bf16, int ldy, int col0)
{
    __shared__ f32x4h part[3][NB][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int groups = (M + 15) >> 4, nkb = K >> 6;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        int row = g * 16 + r;
        row = row < M ? row : M - 1;
        const int b = row / rows_per_img;
        const T *xr = X + (size_t)b * img_stride + (size_t)(row - b * rows_per_img) * ldx + 16 * q;
        // wider heads (COCO: 80 | 81 rows) take several slices of 16 NB columns: X comes from HBM for the first one and from L2 afterwards
        for (int c0 = 0; c0 < N; c0 += 16 * NB) {
            // weight rows of this lane's output columns; columns >= N read row N - 1 and are dropped at the store
            const T *wr[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const int col = c0 + nb * 16 + r;
                wr[nb] = W + (size_t)(col < N ? col : N - 1) * K + 16 * q;
            }
            f32x4h acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; nb++) acc[nb] = (f32x4h){0.f, 0.f, 0.f, 0.f};
            float v[16], vn[16], w[NB][16], wn[NB][16];
            if (wave < nkb) {
                head_load16(xr + wave * 64, v);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) head_load16(wr[nb] + wave * 64, w[nb]);
            }
            for (int kb = wave; kb < nkb; kb += 4) {
                if (kb + 4 < nkb) {
                    head_load16(xr + (kb + 4) * 64, vn);
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) head_load16(wr[nb] + (kb + 4) * 64, wn[nb]);
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], w[nb][i], acc[nb], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    v[i] = vn[i];
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) w[nb][i] = wn[nb][i];
                }
            }
            if (wave > 0) {
#pragma unroll
                for (int nb = 0; nb < NB; nb++) part[wave - 1][nb][lane] = acc[nb];
            }
            __syncthreads();
            if (wave == 0) {
                // C layout: column = lane & 15 (output feature), row = 4 * (lane >> 4) + reg (token)
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const f32x4h t = ((acc[nb] + part[0][nb][lane]) + part[1][nb][lane]) + part[2][nb][lane];
                    const int col = c0 + nb * 16 + r;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int orow = g * 16 + 4 * q + j;
                        if (orow < M && col < N) {
                            float o = t[j];
                            if (round_bf16) o = sizeof(T) == 4 ? (float)(bf16)o : (float)(T)o;      // the operand precision (fp32 operands: bf16)
                            Y[(size_t)orow * ldy + col0 + col] = o;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- backward of the narrow heads (autograd of models/__init__.py:190-204, conv_head.py:38) on the same exact-fp32 instruction ----------
// dX[M, K] (bf16) = dY[M, N] (fp32) W[N, K] (bf16), as the transposed product C'[k][m] = sum_n W[n][k] dY[m][n]: a lane's A operands
// of eight MFMAs come from ONE 16-byte load (lane (i, q): W[4s + q][kc + 8i .. + 7]; MFMA e takes element e, i.e. its row i stands for
// column kc + 8i + e), so that after the n loop lane (c, q) holds, for each j, the eight consecutive columns kc + 32q + 8j .. + 7 of token
// m0 + c: one 16-byte store.  A wave owns 16 tokens x 384 columns; dY is used in fp32 (the padded-GEMM route of round 1 rounded it to bf16).
typedef __bf16 bf16x8h __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float *__restrict__ dY, const bf16 *__restrict__ W, bf16 *__restrict__ dX,
                                                        int M, int N, int K, int halves)
{
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int groups = (M + 15) >> 4;
    const int g = wv / halves, h = wv - g * halves;
    if (g >= groups) return;
    const int m = g * 16 + c, mc = m < M ? m : M - 1;
    const int nsteps = (N + 3) >> 2;
    const int kspan = K / halves;                   // multiple of 128
    for (int kc = h * kspan; kc < (h + 1) * kspan; kc += 128) {
        f32x4h acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = (f32x4h){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s = 0; s < nsteps; s++) {
            const int n = 4 * s + q, nc = n < N ? n : N - 1;
            const float dyv = (n < N && m < M) ? dY[(size_t)mc * N + n] : 0.f;        // 0 for n >= N: the clamped weight row adds nothing
            const bf16x8h wv8 = *reinterpret_cast<const bf16x8h *>(W + (size_t)nc * K + kc + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)wv8[e], dyv, acc[e], 0, 0, 0);
        }
        if (m < M) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bf16x8h o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (bf16)acc[e][j];
                *reinterpret_cast<bf16x8h *>(dX + (size_t)m * K + kc + 32 * q + 8 * j) = o;
            }
        }
    }
}

// dW[N <= 32, K] (fp32) = dY^T X: C[n][k] = sum_m dY[m][n] X[m][k], four token rows per MFMA step.  Lane (c, q) loads X[m0 + q][kc + 8c .. + 7]
// (16 bytes, eight MFMAs' B operands; column mapping as above) and dY[m0 + q][16 nb + i] as A.  A workgroup owns 128 columns and
// 256 token rows (64 per wave, 16 steps); the four waves' partial tiles meet in LDS and are added in wave order, the workgroup's sum goes to
// part[slab][32][K], and head_wgrad_reduce_kernel adds the slabs in order: a fixed summation tree, no atomics.
constexpr int HW_ROWS = 256;
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float *__restrict__ dY, const bf16 *__restrict__ X, float *__restrict__ part,
                                                        int M, int N, int K)
{
    __shared__ f32x4h red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int kc = blockIdx.x * 128, slab = blockIdx.y;
    const int r0 = slab * HW_ROWS + wave * (HW_ROWS / 4);
    f32x4h acc[2][8];
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[nb][e] = (f32x4h){0.f, 0.f, 0.f, 0.f};
    const bool two = N > 16;
#pragma unroll 4
    for (int it = 0; it < HW_ROWS / 16; it++) {
        const int m = r0 + 4 * it + q, mc = m < M ? m : M - 1;
        const bf16x8h xv = *reinterpret_cast<const bf16x8h *>(X + (size_t)mc * K + kc + 8 * c);
        const float a0 = (m < M && c < N) ? dY[(size_t)mc * N + c] : 0.f;
        const float a1 = (m < M && 16 + c < N) ? dY[(size_t)mc * N + 16 + c] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xe = (float)xv[e];
            acc[0][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, xe, acc[0][e], 0, 0, 0);
            if (two) acc[1][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, xe, acc[1][e], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int e = 0; e < 8; e++) red[wave - 1][nb * 8 + e][lane] = acc[nb][e];
    }
    __syncthreads();
    if (wave == 0) {
        float *dst = part + (size_t)slab * 32 * K;
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; e++)
                    o[e] = ((acc[nb][e][j] + red[0][nb * 8 + e][lane][j]) + red[1][nb * 8 + e][lane][j]) + red[2][nb * 8 + e][lane][j];
                float *d = dst + (size_t)(nb * 16 + 4 * q + j) * K + kc + 8 * c;
                *reinterpret_cast<float4 *>(d) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4 *>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
    }
}

__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int N, int K, int slabs)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N * K) return;
    const int n = e / K, k = e - n * K;
    float t = 0.f;
    for (int s = 0; s < slabs; s++) t += part[((size_t)s * 32 + n) * K + k];
    dW[e] = t;
}

// ---- im2col of the stride-16 patch projection for an image batch AND its horizontal flip (seg_helper.py:241-246: every scale goes through
// the teacher as cat(x, x.flip(-1))): cols[(f*B + b)*h*w + py*w + px][c*P*P + dy*P + dx] = x[b][c][P*py + dy][f ? W-1-(P*px+dx) : P*px+dx]
// in the 16-bit operand type.  Replaces flip + cat + .to(16 bit) + the permuted .contiguous() (1.3 GB of traffic per teacher pass).
template <typename T>
__global__ __launch_bounds__(256) void im2col_flip_kernel(const float *__restrict__ x, T *__restrict__ cols, int B, int C, int H, int W, int P, int flips)
{
    typedef T t8 __attribute__((ext_vector_type(8)));
    const int h = H / P, w = W / P, KC = C * P * P, K8 = KC / 8;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)flips * B * h * w * K8;
    if (e >= total) return;
    const int k = (int)(e % K8) * 8;
    size_t row = e / K8;
    const int px = (int)(row % w);
    row /= w;
    const int py = (int)(row % h);
    row /= h;
    const int b = (int)(row % B), f = (int)(row / B);
    const int c = k / (P * P), dy = (k - c * P * P) / P, dx = k % P;          // P % 8 == 0: the 8 elements share (c, dy)
    const float *src = x + (((size_t)b * C + c) * H + (size_t)P * py + dy) * W;
    const int x0 = P * px + dx;
    t8 o;
    if (!f) {
        const float4 a = *reinterpret_cast<const float4 *>(src + x0), d = *reinterpret_cast<const float4 *>(src + x0 + 4);
        o[0] = (T)a.x; o[1] = (T)a.y; o[2] = (T)a.z; o[3] = (T)a.w; o[4] = (T)d.x; o[5] = (T)d.y; o[6] = (T)d.z; o[7] = (T)d.w;
    } else {
        const int xe = W - 8 - x0;                                             // source columns xe .. xe + 7, reversed
        const float4 a = *reinterpret_cast<const float4 *>(src + xe), d = *reinterpret_cast<const float4 *>(src + xe + 4);
        o[0] = (T)d.w; o[1] = (T)d.z; o[2] = (T)d.y; o[3] = (T)d.x; o[4] = (T)a.w; o[5] = (T)a.z; o[6] = (T)a.y; o[7] = (T)a.x;
    }
    reinterpret_cast<t8 *>(cols)[e] = o;
}

// the same rows as fp16c8 operand rows (c8.hpp: hi fp16 | lo8 | hi8 | aug = (1, 1, 0, ...)): the patch projection of the parity-grade teacher
__global__ __launch_bounds__(256) void im2col_flip_c8_kernel(const float *__restrict__ x, unsigned char *__restrict__ rows, int B, int C, int H, int W, int P,
                                                            int flips, int cls_rows)
{
    const int h = H / P, w = W / P, KC = C * P * P, K8 = KC / 8;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)flips * B * h * w * K8;
    if (e >= total) return;
    const int k8 = (int)(e % K8), k = k8 * 8;
    const size_t row_id = e / K8;
    size_t row = row_id;
    const int px = (int)(row % w);
    row /= w;
    const int py = (int)(row % h);
    row /= h;
    const int b = (int)(row % B), f = (int)(row / B);
    const int c = k / (P * P), dy = (k - c * P * P) / P, dx = k % P;
    const float *src = x + (((size_t)b * C + c) * H + (size_t)P * py + dy) * W;
    const int x0 = P * px + dx;
    float v[8];
    if (!f) {
        const float4 a = *reinterpret_cast<const float4 *>(src + x0), d = *reinterpret_cast<const float4 *>(src + x0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
    } else {
        const int xe = W - 8 - x0;
        const float4 a = *reinterpret_cast<const float4 *>(src + xe), d = *reinterpret_cast<const float4 *>(src + xe + 4);
        v[0] = d.w; v[1] = d.z; v[2] = d.y; v[3] = d.x; v[4] = a.w; v[5] = a.z; v[6] = a.y; v[7] = a.x;
    }
    // cls_rows: the output leaves room for that many (zero) rows in front of every image's patch rows -- the class-token slots of the token
    // matrix, so that the patch projection's residual epilogue writes the residual stream in place (cosa_im2col_flip_c8_tokens)
    const size_t out_row = row_id + (size_t)cls_rows * ((size_t)(f * B + b) + 1);
    unsigned char *r = rows + out_row * (size_t)(4 * KC + 128);
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 hi;
    unsigned lo8[2], hi8[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const float vv[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        _Float16 hh[4];
        c8_split4(vv, hh, lo8[q], hi8[q]);
#pragma unroll
        for (int j = 0; j < 4; j++) hi[4 * q + j] = hh[j];
    }
    *reinterpret_cast<h8 *>(r + 2 * k) = hi;
    *reinterpret_cast<uint2 *>(r + 2 * KC + k) = make_uint2(lo8[0], lo8[1]);
    *reinterpret_cast<uint2 *>(r + 3 * KC + k) = make_uint2(hi8[0], hi8[1]);
    if (k8 < 8) *reinterpret_cast<uint4 *>(r + 4 * KC + 16 * k8) = k8 == 0 ? make_uint4(0x3c003c00u, 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
}

// ---- token assembly of the no-grad encoder passes (vit.py:303-313: cat(cls, patch tokens) + pos_embed) straight into the fp32 residual
// stream: out[b][0] = cls + pos[0], out[b][1 + i] = tok[b][i] + pos[1 + i], each sum rounded to the 16-bit operand type first (what the
// 16-bit torch expression did) and then widened.  One pass instead of cat + add + float() + the concatenation of the scales.
template <typename T>
__global__ __launch_bounds__(256) void embed_finish_kernel(const T *__restrict__ tok, const T *__restrict__ cls, const T *__restrict__ pos,
                                                          float *__restrict__ out, int B, int n, int D8)
{
    typedef T t8 __attribute__((ext_vector_type(8)));
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * (n + 1) * D8;
    if (e >= total) return;
    const int c = (int)(e % D8);
    const size_t row = e / D8;
    const int j = (int)(row % (n + 1));
    const size_t b = row / (n + 1);
    const t8 a = j == 0 ? reinterpret_cast<const t8 *>(cls)[c] : reinterpret_cast<const t8 *>(tok)[(b * n + (j - 1)) * D8 + c];
    const t8 p = reinterpret_cast<const t8 *>(pos)[(size_t)j * D8 + c];
    float4 o0, o1;
    o0.x = (float)(T)((float)a[0] + (float)p[0]); o0.y = (float)(T)((float)a[1] + (float)p[1]);
    o0.z = (float)(T)((float)a[2] + (float)p[2]); o0.w = (float)(T)((float)a[3] + (float)p[3]);
    o1.x = (float)(T)((float)a[4] + (float)p[4]); o1.y = (float)(T)((float)a[5] + (float)p[5]);
    o1.z = (float)(T)((float)a[6] + (float)p[6]); o1.w = (float)(T)((float)a[7] + (float)p[7]);
    float4 *d = reinterpret_cast<float4 *>(out + e * 8);
    d[0] = o0;
    d[1] = o1;
}

// ---- GELU' for the training backward of mlp.fc1 (autograd of vit.py:97-98): dH = dA * gelu_erf'(H), bf16, 8 elements per lane -------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16 *__restrict__ dA, const bf16 *__restrict__ H, bf16 *__restrict__ dH, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 g = reinterpret_cast<const bf16x8 *>(dA)[i], h = reinterpret_cast<const bf16x8 *>(H)[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = (float)h[j];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            o[j] = (bf16)((float)g[j] * (cdf + x * pdf));
        }
        reinterpret_cast<bf16x8 *>(dH)[i] = o;
    }
}

// ---- transposed bf16 shadows of the student's projection weights: the input-gradient GEMM dX = dY W is the forward GEMM kernel on W^T --
struct TransposeRec { const float *src; bf16 *dst; int rows, cols, tile0, tiles_c; };       // src [rows, cols] fp32 -> dst [cols, rows] bf16
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TransposeRec *__restrict__ recs, int nrec)
{
    // 64 x 64 tiles; 16-byte loads (a row of the tile = 16 lanes), 4-byte stores (a lane writes two consecutive destination elements).
    // Measured per step (85 M weights): 4-byte loads + 2-byte stores 160 us; 128-row tiles (33 KB of LDS) 254-260 us; this form 112 us (4.5 TB/s)
    __shared__ float tile[64][65];
    typedef __bf16 bf16x2t __attribute__((ext_vector_type(2)));
    int t = 0;
    while (t + 1 < nrec && (int)blockIdx.x >= recs[t + 1].tile0) t++;
    const TransposeRec r = recs[t];
    const int lt = blockIdx.x - r.tile0;
    const int r0 = (lt / r.tiles_c) * 64, c0 = (lt % r.tiles_c) * 64;
    const int tid = threadIdx.x;
    const bool vec = (r.cols & 3) == 0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int i = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4;
        const int rr = r0 + i, cc = c0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < r.rows) {
            if (vec && cc + 3 < r.cols) v = *reinterpret_cast<const float4 *>(r.src + (size_t)rr * r.cols + cc);
            else {
                if (cc < r.cols) v.x = r.src[(size_t)rr * r.cols + cc];
                if (cc + 1 < r.cols) v.y = r.src[(size_t)rr * r.cols + cc + 1];
                if (cc + 2 < r.cols) v.z = r.src[(size_t)rr * r.cols + cc + 2];
                if (cc + 3 < r.cols) v.w = r.src[(size_t)rr * r.cols + cc + 3];
            }
        }
        tile[i][c4] = v.x; tile[i][c4 + 1] = v.y; tile[i][c4 + 2] = v.z; tile[i][c4 + 3] = v.w;
    }
    __syncthreads();
    const bool even = (r.rows & 1) == 0;
    const int l32 = tid & 31, sub = tid >> 5;            // 8 half-waves: each writes one destination row segment (64 elements) per trip
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int i = sub + 8 * it;
        const int cc = c0 + i, rr = r0 + 2 * l32;
        if (cc >= r.cols || rr >= r.rows) continue;
        bf16 *d = r.dst + (size_t)cc * r.rows + rr;
        if (even) {
            *reinterpret_cast<bf16x2t *>(d) = (bf16x2t){(bf16)tile[2 * l32][i], (bf16)tile[2 * l32 + 1][i]};
        } else {
            d[0] = (bf16)tile[2 * l32][i];
            if (rr + 1 < r.rows) d[1] = (bf16)tile[2 * l32 + 1][i];
        }
    }
}

// ---- bicubic resize of the 14 x 14 position grid to the token grid (vit.py:288-291) as a 16-tap gather: the resize is linear in pos_embed
// with <= 16 non-zero weights per output token (the rows of the interpolation matrix), so it is out[p][c] = sum_t w[p][t] * pe[idx[p][t]][c]
__global__ __launch_bounds__(256) void pos_resize_kernel(const float *__restrict__ pe, const int *__restrict__ idx, const float *__restrict__ wgt,
                                                        float *__restrict__ out, int P, int D4)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= P * D4) return;
    const int p = e / D4, c = e - p * D4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const float w = wgt[p * 16 + t];
        const float4 v = reinterpret_cast<const float4 *>(pe + (size_t)idx[p * 16 + t] * D4 * 4)[c];
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    reinterpret_cast<float4 *>(out)[e] = acc;
}

}  // namespace
}  // namespace cosa

using namespace cosa;

extern "C" int cosa_pos_resize(const float *pe, const int *idx, const float *wgt, float *out, int P, int D, void *stream)
{
    COSA_REQUIRE(pe && idx && wgt && out && P > 0 && D > 0 && D % 4 == 0, "cosa_pos_resize: bad arguments (D %% 4 == 0)");
    hipLaunchKernelGGL(pos_resize_kernel, dim3((P * (D / 4) + 255) / 256), dim3(256), 0, as_stream(stream), pe, idx, wgt, out, P, D / 4);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_gelu_backward(const void *dA, const void *H, void *dH, long long n, void *stream)
{
    COSA_REQUIRE(dA && H && dH && n > 0 && n % 8 == 0, "cosa_gelu_backward: n must be a positive multiple of 8");
    const size_t n8 = (size_t)n / 8;
    size_t blocks = (n8 + 255) / 256;
    blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), static_cast<const bf16 *>(dA),
                       static_cast<const bf16 *>(H), static_cast<bf16 *>(dH), n8);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" size_t cosa_transpose_record_bytes(void) { return sizeof(TransposeRec); }
/* records: device array of n {const float *src; bf16 *dst; int rows, cols, tile0, tiles_c}: dst[cols, rows] = bf16(src[rows, cols]^T);
 * tile0 = number of 64x64 tiles of the records before this one, tiles_c = ceil(cols / 64); total_tiles = the grand total          */
extern "C" int cosa_transpose_cast_batched(const void *records, int n, int total_tiles, void *stream)
{
    COSA_REQUIRE(records && n > 0 && total_tiles > 0, "cosa_transpose_cast_batched: bad arguments");
    hipLaunchKernelGGL(transpose_cast_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), static_cast<const TransposeRec *>(records), n);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// Y[M,N] fp32 = X W^T for narrow N (CAM / seg heads: 20 | 21 | 80 | 81 rows); X rows: image b = rows [b*rows_per_img, +rows_per_img) at X + b*img_stride
// (elements) with row stride ldx, so token views without their cls row need no copy.  dtype: 0 = fp32 operands, 1 = bf16
// operands (round_bf16 = 1 additionally rounds the result to bf16 precision, like a bf16 library GEMM would).  The N columns land
// at Y[r*ldy + col0 ...]; wider heads (COCO: 80 | 81 rows) run as slices of 32 weight rows inside the kernel (X is then re-read from L2).
extern "C" int cosa_head_gemm(const void *X, const void *W, float *Y, int M, int N, int K, int rows_per_img, long long img_stride,
                              int ldx, int dtype, int round_bf16, int ldy, int col0, void *stream)
{
    COSA_REQUIRE(ldy >= col0 + N && col0 >= 0, "cosa_head_gemm: output columns [col0, col0+N) must fit the row stride ldy");
    COSA_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0 && rows_per_img > 0, "cosa_head_gemm: bad arguments");
    COSA_REQUIRE(N <= 1024 && K % 64 == 0 && ldx >= K, "cosa_head_gemm: N <= 1024 and K %% 64 == 0 (got N=%d K=%d)", N, K);
    COSA_REQUIRE(dtype >= 0 && dtype <= 2, "cosa_head_gemm: dtype 0 (fp32), 1 (bf16) or 2 (fp16)");
    COSA_REQUIRE(ldx % (dtype == 0 ? 4 : 8) == 0 && img_stride % (dtype == 0 ? 4 : 8) == 0, "cosa_head_gemm: rows must be 16-byte aligned");
    const int nb = N <= 16 ? 1 : 2;
    hipStream_t st = as_stream(stream);
    int blocks = (M + 15) / 16;
    blocks = blocks > 8192 ? 8192 : blocks;
#define COSA_HEAD_LAUNCH(T, NB)                                                                                                   \
    hipLaunchKernelGGL((head_gemm_kernel<T, NB>), dim3(blocks), dim3(256), 0, st, static_cast<const T *>(X), static_cast<const T *>(W), Y, M, N, \
                       K, rows_per_img, img_stride, ldx, round_bf16, ldy, col0)
    if (dtype == 0) { if (nb == 1) COSA_HEAD_LAUNCH(float, 1); else COSA_HEAD_LAUNCH(float, 2); }
    else if (dtype == 1) { if (nb == 1) COSA_HEAD_LAUNCH(bf16, 1); else COSA_HEAD_LAUNCH(bf16, 2); }
    else { if (nb == 1) COSA_HEAD_LAUNCH(_Float16, 1); else COSA_HEAD_LAUNCH(_Float16, 2); }
#undef COSA_HEAD_LAUNCH
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// backward of the narrow heads: dX [M,K] bf16 = dY [M,N] (fp32, contiguous) W [N,K] (bf16)   (any N, K % 128 == 0)
extern "C" int cosa_head_gemm_dgrad(const float *dY, const void *W, void *dX, int M, int N, int K, void *stream)
{
    COSA_REQUIRE(dY && W && dX && M > 0 && N > 0 && K > 0 && K % 128 == 0, "cosa_head_gemm_dgrad: K %% 128 == 0 (got N=%d K=%d)", N, K);
    const int halves = (K % 256 == 0) ? 2 : 1;
    const long waves = (long)((M + 15) / 16) * halves;
    hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), dY, static_cast<const bf16 *>(W),
                       static_cast<bf16 *>(dX), M, N, K, halves);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" size_t cosa_head_gemm_wgrad_workspace(int M, int K) { return (size_t)((M + HW_ROWS - 1) / HW_ROWS) * 32 * (size_t)K * sizeof(float); }
// dW [N,K] fp32 = dY^T [N,M] X [M,K] (X bf16, contiguous);  workspace: cosa_head_gemm_wgrad_workspace(M, K) bytes
extern "C" int cosa_head_gemm_wgrad(const float *dY, const void *X, float *dW, void *workspace, int M, int N, int K, void *stream)
{
    COSA_REQUIRE(dY && X && dW && workspace && M > 0 && N > 0 && N <= 32 && K > 0 && K % 128 == 0, "cosa_head_gemm_wgrad: N <= 32 and K %% 128 == 0 (got N=%d K=%d)", N, K);
    const int slabs = (M + HW_ROWS - 1) / HW_ROWS;
    COSA_REQUIRE(slabs <= 65535, "cosa_head_gemm_wgrad: too many rows");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(K / 128, slabs), dim3(256), 0, st, dY, static_cast<const bf16 *>(X), static_cast<float *>(workspace), M, N, K);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((N * K + 255) / 256), dim3(256), 0, st, static_cast<const float *>(workspace), dW, N, K, slabs);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// cols [flips * B * (H/P) * (W/P), C*P*P] (16-bit, dtype 1 = bf16 / 2 = fp16) = im2col of x [B,C,H,W] fp32 (flips = 1) or of cat(x, x.flip(-1)) (flips = 2)
extern "C" int cosa_im2col_flip(const float *x, void *cols, int B, int C, int H, int W, int P, int flips, int dtype, void *stream)
{
    COSA_REQUIRE(x && cols && B > 0 && C > 0 && H > 0 && W > 0 && P > 0, "cosa_im2col_flip: bad arguments");
    COSA_REQUIRE(P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "cosa_im2col_flip: patch size must be a multiple of 8 and divide H and W (got P=%d H=%d W=%d)", P, H, W);
    COSA_REQUIRE((flips == 1 || flips == 2) && (dtype >= 1 && dtype <= 3), "cosa_im2col_flip: flips 1 | 2, dtype 1 (bf16) | 2 (fp16) | 3 (fp16c8 rows)");
    COSA_REQUIRE(dtype != 3 || (C * P * P) % 128 == 0, "cosa_im2col_flip: fp16c8 rows need C*P*P %% 128 == 0");
    const size_t total = (size_t)flips * B * (H / P) * (W / P) * (C * P * P / 8);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == 3)
        hipLaunchKernelGGL(im2col_flip_c8_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, static_cast<unsigned char *>(cols), B, C, H, W, P, flips, 0);
    else if (dtype == 1)
        hipLaunchKernelGGL(im2col_flip_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), x, static_cast<bf16 *>(cols), B, C, H, W, P, flips);
    else
        hipLaunchKernelGGL(im2col_flip_kernel<_Float16>, dim3(grid), dim3(256), 0, as_stream(stream), x, static_cast<_Float16 *>(cols), B, C, H, W, P, flips);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// fp16c8 rows of the im2col view of cat(x, x.flip(-1)) (flips = 2) written INTO a token matrix: image (f, b)'s patch rows start at row
// (f B + b) (n + cls_rows) + cls_rows, n = (H/P)(W/P); the cls_rows rows in front of them are not touched (the caller keeps them zero, incl. the
// augmentation block, so the patch GEMM adds nothing there, not even the bias)
extern "C" int cosa_im2col_flip_c8_tokens(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream)
{
    COSA_REQUIRE(x && rows && B > 0 && C > 0 && H > 0 && W > 0 && P > 0 && cls_rows >= 0, "cosa_im2col_flip_c8_tokens: bad arguments");
    COSA_REQUIRE(P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "cosa_im2col_flip_c8_tokens: patch size must be a multiple of 8 and divide H and W");
    COSA_REQUIRE((flips == 1 || flips == 2) && (C * P * P) % 128 == 0, "cosa_im2col_flip_c8_tokens: flips 1 | 2, C*P*P %% 128 == 0");
    const size_t total = (size_t)flips * B * (H / P) * (W / P) * (C * P * P / 8);
    hipLaunchKernelGGL(im2col_flip_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, static_cast<unsigned char *>(rows), B, C, H, W, P,
                       flips, cls_rows);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// dst [B][n] fp32 = src [n] for every b: the fp32 residual stream of a no-grad pass starts as (cls + pos_0 | pos rows) per image before the patch
// projection adds into it in place (models/vit/vit.py:283-300); n % 4 == 0
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int B, size_t n4)
{
    const size_t total = (size_t)B * n4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) dst[e] = src[e % n4];
}

extern "C" int cosa_broadcast_rows(const float *src, float *dst, int B, long long n, void *stream)
{
    COSA_REQUIRE(src && dst && B > 0 && n > 0 && n % 4 == 0, "cosa_broadcast_rows: bad arguments (n %% 4 == 0)");
    const size_t total = (size_t)B * (size_t)(n / 4);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<float4 *>(dst), B, (size_t)(n / 4));
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// out [B, n+1, D] fp32 = (cls | tok [B, n, D]) + pos [n+1, D]; tok / cls / pos in one 16-bit type (dtype 1 = bf16, 2 = fp16), D % 8 == 0
extern "C" int cosa_embed_finish(const void *tok, const void *cls, const void *pos, float *out, int B, int n, int D, int dtype, void *stream)
{
    COSA_REQUIRE(tok && cls && pos && out && B > 0 && n > 0 && D > 0 && D % 8 == 0, "cosa_embed_finish: bad arguments (D %% 8 == 0)");
    COSA_REQUIRE(dtype == 1 || dtype == 2, "cosa_embed_finish: dtype 1 (bf16) or 2 (fp16)");
    const size_t total = (size_t)B * (n + 1) * (D / 8);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == 1)
        hipLaunchKernelGGL(embed_finish_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), static_cast<const bf16 *>(tok),
                           static_cast<const bf16 *>(cls), static_cast<const bf16 *>(pos), out, B, n, D / 8);
    else
        hipLaunchKernelGGL(embed_finish_kernel<_Float16>, dim3(grid), dim3(256), 0, as_stream(stream), static_cast<const _Float16 *>(tok),
                           static_cast<const _Float16 *>(cls), static_cast<const _Float16 *>(pos), out, B, n, D / 8);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_add_layernorm_fwd(const void *x, const void *delta, const void *gamma, const void *beta, void *x_out, void *y,
                                      float *mean, float *rstd, int rows, int dim, float eps, void *stream)
{
    COSA_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0, "cosa_add_layernorm_fwd: bad arguments");
    COSA_REQUIRE(dim == D, "cosa_add_layernorm_fwd: dim must be 768 (ViT-B)");
    COSA_REQUIRE(!delta || x_out, "cosa_add_layernorm_fwd: x_out is required with delta");
    hipLaunchKernelGGL(add_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), static_cast<const bf16 *>(x),
                       static_cast<const bf16 *>(delta), static_cast<const bf16 *>(gamma), static_cast<const bf16 *>(beta),
                       static_cast<bf16 *>(x_out), static_cast<bf16 *>(y), mean, rstd, rows, eps);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" size_t cosa_layernorm_bwd_workspace_bytes(int rows, int dim)
{
    (void)rows;
    return (size_t)1024 * 2 * dim * sizeof(float);             // partial rows of up to 1024 workgroups
}

extern "C" int cosa_layernorm_bwd(const void *dy, const void *x_new, const float *mean, const float *rstd, const void *gamma,
                                  const void *dskip, void *dx, float *dgamma, float *dbeta, int accumulate, int rows, int dim,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(dy && x_new && mean && rstd && gamma && dx && dgamma && dbeta && workspace && rows > 0, "cosa_layernorm_bwd: bad arguments");
    COSA_REQUIRE(dim == D, "cosa_layernorm_bwd: dim must be 768 (ViT-B)");
    COSA_REQUIRE(workspace_bytes >= cosa_layernorm_bwd_workspace_bytes(rows, dim), "cosa_layernorm_bwd: workspace too small");
    // two workgroups per CU: a wave walks its rows serially with one row of loads in flight, so the kernel is latency-bound on occupancy
    // (256 / 512 / 768 / 1024 workgroups: 22.3 + 4.8 / 16.0 + 6.5 / 17.1 + 8.1 / 18.9 + 9.5 us for the kernel + its partial-sum reduction)
    constexpr int target_blocks = 512;
    int per = (rows + target_blocks - 1) / target_blocks;
    per = (per + 3) / 4 * 4;                                  // whole rounds of the workgroup's 4 waves
    const int nblk = (rows + per - 1) / per;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), 0, st, static_cast<const bf16 *>(dy), static_cast<const bf16 *>(x_new), mean,
                       rstd, static_cast<const bf16 *>(gamma), static_cast<const bf16 *>(dskip), static_cast<bf16 *>(dx),
                       static_cast<float *>(workspace), rows, per);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(2 * D / 8), dim3(256), 0, st, static_cast<const float *>(workspace), nblk, dgamma, dbeta,
                       accumulate);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// Junction of the training path's heads (models/__init__.py:163-206 + autograd): decoder, CAM head and pooled classification head all read the
// patch tokens (the class token feeds nothing there) of ONE bf16 copy of the fp32 final tokens; their bf16 gradients [B, N - 1, D] meet here:
// dx [B, N, D] fp32 = 0 in the class-token row, g0 + g1 + g2 (fp32 adds in that order; absent consumers are null) elsewhere -- one pass
// instead of autograd's zero-fill + slice copy per consumer, a cast and the adds (round 4: ~20 ATen launches per step).
__global__ __launch_bounds__(256) void token_junction_bwd_kernel(const bf16 *__restrict__ g0, const bf16 *__restrict__ g1, const bf16 *__restrict__ g2,
                                                                float *__restrict__ dx, int N, long long total4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;            // one float4 of dx
    if (i >= total4) return;
    constexpr int D4 = D / 4;
    const long long row = i / D4;
    const int c4 = (int)(i - row * D4), t = (int)(row % N);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t != 0) {
        const long long b = row / N;
        const size_t src = ((size_t)(b * (N - 1) + t - 1) * D4 + c4) * 4;
        auto ld = [&](const bf16 *g, float (&v)[4]) {
            const uint2 u = *reinterpret_cast<const uint2 *>(g + src);
            v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
            v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
        };
        float a[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
        bool first = true;
        for (const bf16 *g : {g0, g1, g2}) {
            if (!g) continue;
            ld(g, v);
            if (first) { a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3]; first = false; }
            else { a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3]; }
        }
        o = make_float4(a[0], a[1], a[2], a[3]);
    }
    *reinterpret_cast<float4 *>(dx + i * 4) = o;
}

extern "C" int cosa_token_junction_bwd(const void *g0, const void *g1, const void *g2, float *dx, int B, int N, int dim, void *stream)
{
    COSA_REQUIRE((g0 || g1 || g2) && dx && B > 0 && N > 1, "cosa_token_junction_bwd: bad arguments");
    COSA_REQUIRE(dim == D, "cosa_token_junction_bwd: dim must be 768 (ViT-B)");
    const long long total4 = (long long)B * N * (D / 4);
    hipLaunchKernelGGL(token_junction_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const bf16 *>(g0),
                       static_cast<const bf16 *>(g1), static_cast<const bf16 *>(g2), dx, N, total4);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_layernorm_bwd_f32(const void *dy, int dy_is_f32, const float *x, const void *gamma, const float *dskip, float *dx, void *dx16,
                                      float *dgamma, float *dbeta, int accumulate, int rows, int dim, float eps, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && workspace && rows > 0, "cosa_layernorm_bwd_f32: bad arguments");
    COSA_REQUIRE(dim == D, "cosa_layernorm_bwd_f32: dim must be 768 (ViT-B)");
    COSA_REQUIRE(workspace_bytes >= cosa_layernorm_bwd_workspace_bytes(rows, dim), "cosa_layernorm_bwd_f32: workspace too small");
    constexpr int target_blocks = 512;
    int per = (rows + target_blocks - 1) / target_blocks;
    per = (per + 3) / 4 * 4;
    const int nblk = (rows + per - 1) / per;
    hipStream_t st = as_stream(stream);
    if (dy_is_f32)
        hipLaunchKernelGGL(ln_bwd_f32_kernel<true>, dim3(nblk), dim3(256), 0, st, dy, x, static_cast<const bf16 *>(gamma), dskip,
                           dx, static_cast<bf16 *>(dx16), static_cast<float *>(workspace), rows, per, eps);
    else
        hipLaunchKernelGGL(ln_bwd_f32_kernel<false>, dim3(nblk), dim3(256), 0, st, dy, x, static_cast<const bf16 *>(gamma), dskip,
                           dx, static_cast<bf16 *>(dx16), static_cast<float *>(workspace), rows, per, eps);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(2 * D / 8), dim3(256), 0, st, static_cast<const float *>(workspace), nblk, dgamma, dbeta,
                       accumulate);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
