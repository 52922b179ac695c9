// split_kernels.hip -- producers of bf16x3 ("split") operands for the parity-grade no-grad passes (gfx950, HBM-bound).
//
// A value v is carried as hi = bf16(v) and lo = bf16(v - hi): 16 significant bits.  A split operand row of logical width K is
//     [ hi (K) | lo (K) | aug (64) ]           row stride 2K + 64 bf16
// (layout and the three-term product: gemm_kernels.hip, split_tile_x / split_tile_w).  The augmentation block of an ACTIVATION row is
// (1, 1, 0, ...), that of a WEIGHT row n is (bias_hi[n], bias_lo[n], 0, ...): the GEMM adds the bias at 16-bit precision as one more K
// tile, so nn.Linear's bias (models/vit/vit.py:96-102,121,135) needs no epilogue work.
//
//   cosa_split_rows        fp32 [R, K] (+ optional fp32 bias [R]) -> split rows (weights: called once per step for the EMA-updated
//                          teacher; activations: the im2col'd image of the patch projection)
//   cosa_layernorm_split   nn.LayerNorm(768, eps) over the fp32 residual stream with fp32 gamma / beta -> split rows (and/or fp32)
#include "kernels.hpp"
#include "c8.hpp"
#include "c4.hpp"

namespace cosa {
namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// T: the 16-bit type of the halves -- __bf16 (bf16x3: 8 + 8 significant bits, fp32's exponent range) or _Float16 (fp16x3, round 6: 11 + 11 bits;
// a lo half below 2^-14 is an fp16 subnormal, i.e. carries an absolute 2^-24: still >= 18 significant bits for |v| >= 2^-6)
template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));
template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ void split4(const float (&v)[4], vec4<T> &hi, vec4<T> &lo)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hi[j] = (T)v[j];
        lo[j] = (T)(v[j] - (float)hi[j]);
    }
}

// one wave per row; K % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ src, const float *__restrict__ bias, T *__restrict__ dst,
                                                        int R, int K, long long src_ld, int ones)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float *s = src + (size_t)row * src_ld;
    T *d = dst + (size_t)row * (2 * K + 64);
    for (int c = lane * 4; c < K; c += 256) {
        const float4 f = *reinterpret_cast<const float4 *>(s + c);
        const float v[4] = {f.x, f.y, f.z, f.w};
        vec4<T> hi, lo;
        split4<T>(v, hi, lo);
        *reinterpret_cast<vec4<T> *>(d + c) = hi;
        *reinterpret_cast<vec4<T> *>(d + K + c) = lo;
    }
    if (lane < 8) {                                   // augmentation block: 8 lanes x 8 halves
        vec8<T> a;
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (T)0.f;
        if (lane == 0) {
            if (ones) { a[0] = (T)1.f; a[1] = (T)1.f; }
            else if (bias) { const float b = bias[row]; a[0] = (T)b; a[1] = (T)(b - (float)a[0]); }
        }
        *reinterpret_cast<vec8<T> *>(d + 2 * K + lane * 8) = a;
    }
}

// ---- im2col of the stride-16 patch projection for an image batch and (flips = 2) its horizontal flip, written as SPLIT rows INTO a token
// matrix (the x3 analogue of vit_kernels.hip: im2col_flip_c8_kernel): the row of patch (f, b, py, px) is
//     cols[c*P*P + dy*P + dx] = x[b][c][P*py + dy][f ? W-1-(P*px+dx) : P*px+dx]          as  [hi | lo | aug = (1, 1, 0, ...)]
// at token row (f*B + b) * (h*w + cls_rows) + cls_rows + py*w + px: every image's patch rows leave `cls_rows` rows in front of them untouched --
// the class-token slots of the token matrix, zero for good (augmentation block included: no bias there) -- so that ONE patch projection over
// the tokens of all scales adds into the fp32 residual stream in place.  Replaces flip + cat, the permuted .contiguous(), split_rows, the
// position-row broadcast and two concatenations per scale of the x3 teacher (0.75 ms per step in ATen kernels, round 6).
template <typename T>
__global__ __launch_bounds__(256) void im2col_flip_split_kernel(const float *__restrict__ x, T *__restrict__ rows, int B, int C, int H, int W, int P,
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
    const int c = k / (P * P), dy = (k - c * P * P) / P, dx = k % P;          // P % 8 == 0: the 8 elements share (c, dy)
    const float *src = x + (((size_t)b * C + c) * H + (size_t)P * py + dy) * W;
    const int x0 = P * px + dx;
    float v0[4], v1[4];
    if (!f) {
        const float4 a = *reinterpret_cast<const float4 *>(src + x0), d = *reinterpret_cast<const float4 *>(src + x0 + 4);
        v0[0] = a.x; v0[1] = a.y; v0[2] = a.z; v0[3] = a.w; v1[0] = d.x; v1[1] = d.y; v1[2] = d.z; v1[3] = d.w;
    } else {
        const int xe = W - 8 - x0;                                             // source columns xe .. xe + 7, reversed
        const float4 a = *reinterpret_cast<const float4 *>(src + xe), d = *reinterpret_cast<const float4 *>(src + xe + 4);
        v0[0] = d.w; v0[1] = d.z; v0[2] = d.y; v0[3] = d.x; v1[0] = a.w; v1[1] = a.z; v1[2] = a.y; v1[3] = a.x;
    }
    const size_t out_row = row_id + (size_t)cls_rows * ((size_t)(f * B + b) + 1);
    T *r = rows + out_row * (size_t)(2 * KC + 64);
    vec4<T> h0, l0, h1, l1;
    split4<T>(v0, h0, l0);
    split4<T>(v1, h1, l1);
    *reinterpret_cast<vec4<T> *>(r + k) = h0;
    *reinterpret_cast<vec4<T> *>(r + k + 4) = h1;
    *reinterpret_cast<vec4<T> *>(r + KC + k) = l0;
    *reinterpret_cast<vec4<T> *>(r + KC + k + 4) = l1;
    if (k8 < 8) {                                     // augmentation block: 8 threads x 8 halves
        vec8<T> a;
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (T)0.f;
        if (k8 == 0) { a[0] = (T)1.f; a[1] = (T)1.f; }
        *reinterpret_cast<vec8<T> *>(r + 2 * KC + 8 * k8) = a;
    }
}

template <int D, typename T>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b,
                                                             T *__restrict__ y, float *__restrict__ y32, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int PER = D / 64 / 4;
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
    T *yr = y ? y + (size_t)row * (2 * D + 64) : nullptr;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const float4 gg = *reinterpret_cast<const float4 *>(g + c0), bb = *reinterpret_cast<const float4 *>(b + c0);
        const float o[4] = {(v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                            (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w};
        if (yr) {
            vec4<T> hi, lo;
            split4<T>(o, hi, lo);
            *reinterpret_cast<vec4<T> *>(yr + c0) = hi;
            *reinterpret_cast<vec4<T> *>(yr + D + c0) = lo;
        }
        if (y32) *reinterpret_cast<float4 *>(y32 + (size_t)row * D + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (yr && lane < 8) {
        vec8<T> a;
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (T)0.f;
        if (lane == 0) { a[0] = (T)1.f; a[1] = (T)1.f; }
        *reinterpret_cast<vec8<T> *>(yr + 2 * D + lane * 8) = a;
    }
}


// ---- fp16c8 rows (c8.hpp): [hi fp16 (2K bytes) | lo8 (K) | hi8 (K) | aug fp16 (128)] -----------------------------------------------------
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the fp32 product, rounded -- opaque to the optimizer, which otherwise folds the multiply into the fp16 conversion that follows
// (v_fma_mix: one rounding) and into the subtraction of the split (fma): the rows must be those of the pre-multiplied matrix
__device__ __forceinline__ float mul_rounded(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void c8_store4(unsigned char *row, int K, int c, const float (&v)[4])
{
    f16 hi[4];
    unsigned lo8, hi8;
    c8_split4(v, hi, lo8, hi8);
    *reinterpret_cast<f16x4 *>(row + 2 * c) = (f16x4){hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<unsigned *>(row + 2 * K + c) = lo8;
    *reinterpret_cast<unsigned *>(row + 3 * K + c) = hi8;
}

__device__ __forceinline__ void c8_store_aug(unsigned char *row, int K, int lane, bool ones, const float *bias_of_row, float bias_factor = 1.0f)
{
    if (lane < 8) {                                   // augmentation block: 8 lanes x 8 fp16
        f16x8 a;
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = (f16)0.f;
        if (lane == 0) {
            if (ones) { a[0] = (f16)1.f; a[1] = (f16)1.f; }
            else if (bias_of_row) { const float b = mul_rounded(*bias_of_row, bias_factor); a[0] = (f16)b; a[1] = (f16)(b - (float)a[0]); }
        }
        *reinterpret_cast<f16x8 *>(row + 4 * K + lane * 16) = a;
    }
}

// one wave per row; K % 4 == 0
__global__ __launch_bounds__(256) void c8_rows_kernel(const float *__restrict__ src, const float *__restrict__ bias, unsigned char *__restrict__ dst,
                                                     int R, int K, long long src_ld, int ones)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    const float *s = src + (size_t)row * src_ld;
    unsigned char *d = dst + (size_t)row * (4 * K + 128);
    for (int c = lane * 4; c < K; c += 256) {
        const float4 f = *reinterpret_cast<const float4 *>(s + c);
        const float v[4] = {f.x, f.y, f.z, f.w};
        c8_store4(d, K, c, v);
    }
    c8_store_aug(d, K, lane, ones != 0, bias ? bias + row : nullptr);
}

// every weight matrix of a network in ONE launch (the teacher's c8 weight rows are rebuilt from the fp32 masters once per pass: 37 small
// launches of ~9 us otherwise); a record per matrix, rows dealt to waves across all of them
// qrows: the first qrows rows (and their bias entries) are multiplied by kAttnPrescale before the conversion -- the q third of a qkv
// projection whose attention kernel takes q pre-scaled (attn_kernels.hip: AUGM): folded into the weights, the factor costs no rounding
constexpr float kAttnPrescale = 0.125f * 1.4426950408889634f;        // head_dim^-0.5 log2(e), head_dim = 64
struct C8Rec {
    const float *src;
    const float *bias;
    unsigned char *dst;
    int rows, K, row0, qrows;
};

__global__ __launch_bounds__(256) void c8_rows_batched_kernel(const C8Rec *__restrict__ recs, int nrec, int total_rows)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int lane = threadIdx.x & 63;
    int r = 0;
    while (r + 1 < nrec && recs[r + 1].row0 <= row) r++;
    const C8Rec rec = recs[r];
    const int lr = row - rec.row0, K = rec.K;
    const float *s = rec.src + (size_t)lr * K;
    unsigned char *d = rec.dst + (size_t)lr * (4 * K + 128);
    const float f0 = lr < rec.qrows ? kAttnPrescale : 1.0f;
    for (int c = lane * 4; c < K; c += 256) {
        const float4 f = *reinterpret_cast<const float4 *>(s + c);
        const float v[4] = {mul_rounded(f.x, f0), mul_rounded(f.y, f0), mul_rounded(f.z, f0), mul_rounded(f.w, f0)};      // (rounded products: not to be fused into the split)
        c8_store4(d, K, c, v);
    }
    c8_store_aug(d, K, lane, false, rec.bias ? rec.bias + lr : nullptr, f0);
}

template <int D>
__global__ __launch_bounds__(256) void layernorm_c8_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b,
                                                          unsigned char *__restrict__ y, float *__restrict__ y32, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int PER = D / 64 / 4;
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
    unsigned char *yr = y ? y + (size_t)row * (4 * D + 128) : nullptr;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const float4 gg = *reinterpret_cast<const float4 *>(g + c0), bb = *reinterpret_cast<const float4 *>(b + c0);
        const float o[4] = {(v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                            (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w};
        if (yr) c8_store4(yr, D, c0, o);
        if (y32) *reinterpret_cast<float4 *>(y32 + (size_t)row * D + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (yr) c8_store_aug(yr, D, lane, true, nullptr);
}

// ---- fp16c4 rows (c4.hpp): [hi fp16 (2K bytes) | c4 blocks (K) | unused (K) | aug fp16 (128)] + a scale tensor ---------------------------
// one 16-feature block per lane and trip: 64 contiguous source bytes in, 32 bytes of fp16 + one 16-byte block + one scale byte out
__device__ __forceinline__ void c4_store_block(unsigned char *row, unsigned char *scales, int K, int r, int b, const float (&v)[16], bool weight)
{
    f16 hi[16];
    unsigned blk[4];
    int e;
    c4_block16(v, hi, blk, e, weight);
    f16x8 h0, h1;
#pragma unroll
    for (int j = 0; j < 8; j++) { h0[j] = hi[j]; h1[j] = hi[8 + j]; }
    *reinterpret_cast<f16x8 *>(row + 32 * b) = h0;
    *reinterpret_cast<f16x8 *>(row + 32 * b + 16) = h1;
    *reinterpret_cast<uint4 *>(row + 2 * K + 16 * b) = make_uint4(blk[0], blk[1], blk[2], blk[3]);
    const int Kq = K >> 7;
    scales[weight ? c4_scale_off_w(r, b >> 3, b & 7, Kq) : c4_scale_off_x(r, b >> 3, b & 7, Kq)] =
        (unsigned char)c4_scale_byte(e, weight ? kC4WeightExpBias : 0);
}

__device__ __forceinline__ void c4_row_from_f32(const float *s, unsigned char *d, unsigned char *scales, int K, int r, int lane, bool weight,
                                                float factor = 1.0f)
{
    for (int b = lane; b < (K >> 4); b += 64) {
        float v[16];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 f = *reinterpret_cast<const float4 *>(s + 16 * b + 4 * c);
            v[4 * c] = mul_rounded(f.x, factor); v[4 * c + 1] = mul_rounded(f.y, factor); v[4 * c + 2] = mul_rounded(f.z, factor); v[4 * c + 3] = mul_rounded(f.w, factor);
        }
        c4_store_block(d, scales, K, r, b, v, weight);
    }
}

// one wave per row; K % 256 == 0
__global__ __launch_bounds__(256) void c4_rows_kernel(const float *__restrict__ src, const float *__restrict__ bias, unsigned char *__restrict__ dst,
                                                     unsigned char *__restrict__ scales, int R, int K, long long src_ld, int ones, int weight)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    unsigned char *d = dst + (size_t)row * (4 * K + 128);
    c4_row_from_f32(src + (size_t)row * src_ld, d, scales, K, row, lane, weight != 0);
    c8_store_aug(d, K, lane, ones != 0, bias ? bias + row : nullptr);
}

struct C4Rec {
    const float *src;
    const float *bias;
    unsigned char *dst;
    unsigned char *scales;
    int rows, K, row0, qrows;          // qrows: as in C8Rec
};

// the weight matrices of a network in one launch (weight block order, 2^-11 in the scale bytes, bias in the augmentation block)
__global__ __launch_bounds__(256) void c4_rows_batched_kernel(const C4Rec *__restrict__ recs, int nrec, int total_rows)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int lane = threadIdx.x & 63;
    int r = 0;
    while (r + 1 < nrec && recs[r + 1].row0 <= row) r++;
    const C4Rec rec = recs[r];
    const int lr = row - rec.row0, K = rec.K;
    unsigned char *d = rec.dst + (size_t)lr * (4 * K + 128);
    const float f0 = lr < rec.qrows ? kAttnPrescale : 1.0f;
    c4_row_from_f32(rec.src + (size_t)lr * K, d, rec.scales, K, lr, lane, true, f0);
    c8_store_aug(d, K, lane, false, rec.bias ? rec.bias + lr : nullptr, f0);
}

// LayerNorm(768) over the fp32 stream -> c4 rows (activation block order).  The row sits in registers as float4 (lane + 64 i): a 16-feature
// block is the float4s of four consecutive lanes, so the block's amax meets in the quad by two DPP quad permutes, each lane converts its own
// four values with the shared scale and stores 8 bytes of fp16, 2 + 2 bytes of the block and (lane 0 of the quad) the scale byte.
template <int D>
__global__ __launch_bounds__(256) void layernorm_c4_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b,
                                                          unsigned char *__restrict__ y, unsigned char *__restrict__ scales,
                                                          float *__restrict__ y32, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    constexpr int PER = D / 64 / 4;
    constexpr int Kq = D / 128;
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
    unsigned char *yr = y ? y + (size_t)row * (4 * D + 128) : nullptr;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c0 = (lane + 64 * i) * 4;
        const float4 gg = *reinterpret_cast<const float4 *>(g + c0), bb = *reinterpret_cast<const float4 *>(b + c0);
        const float o[4] = {(v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                            (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w};
        if (y32) *reinterpret_cast<float4 *>(y32 + (size_t)row * D + c0) = make_float4(o[0], o[1], o[2], o[3]);
        if (!yr) continue;
        f16 hi[4];
        float h[4], l[4];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hi[j] = (f16)o[j];
            h[j] = (float)hi[j];
            l[j] = (o[j] - h[j]) * kC4LoScale;
            amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(h[j]), __builtin_fabsf(l[j])));
        }
        // quad maximum: lanes 4k .. 4k+3 hold the block's 16 features
        int ai = __builtin_bit_cast(int, amax);          // (non-negative floats order like their bit patterns)
        int t = __builtin_amdgcn_update_dpp(0, ai, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
        ai = ai > t ? ai : t;
        t = __builtin_amdgcn_update_dpp(0, ai, 0x4E, 0xf, 0xf, false);          // quad_perm [2,3,0,1]
        ai = ai > t ? ai : t;
        const int e = c4_block_exp(__builtin_bit_cast(float, ai));
        const float sc = c4_pow2(e);
        unsigned pl = 0, ph = 0;
        pl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl, l[0], l[1], sc, 0);
        pl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(pl, l[2], l[3], sc, 1);
        ph = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph, h[0], h[1], sc, 0);
        ph = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ph, h[2], h[3], sc, 1);
        *reinterpret_cast<f16x4 *>(yr + 2 * c0) = (f16x4){hi[0], hi[1], hi[2], hi[3]};
        const int blk = c0 >> 4, sub = lane & 3;                                 // block index in the row, the lane's place in its quad
        unsigned char *bp = yr + 2 * D + 16 * blk;
        *reinterpret_cast<unsigned short *>(bp + 2 * sub) = (unsigned short)pl;          // lo' half: bytes [0, 8)
        *reinterpret_cast<unsigned short *>(bp + 8 + 2 * sub) = (unsigned short)ph;      // hi half:  bytes [8, 16)
        if (sub == 0) scales[c4_scale_off_x(row, blk >> 3, blk & 7, Kq)] = (unsigned char)c4_scale_byte(e, 0);
    }
    if (yr) c8_store_aug(yr, D, lane, true, nullptr);
}

}  // namespace
}  // namespace cosa

using namespace cosa;

// ---- fp16c4 producers (c4.hpp; include/cosa_hip.h) ---------------------------------------------------------------------------------
extern "C" size_t cosa_c4_scale_bytes(int rows, int K) { return rows > 0 && K > 0 ? (size_t)((rows + 255) / 256) * (size_t)(K / 128) * 2048 : 0; }

extern "C" int cosa_c4_rows(const float *src, const float *bias, void *dst, void *scales, int R, int K, long long src_ld, int ones, int weight,
                            void *stream)
{
    COSA_REQUIRE(src && dst && scales && R > 0 && K > 0 && K % 256 == 0 && src_ld >= K && src_ld % 4 == 0, "cosa_c4_rows: bad arguments (K %% 256 == 0)");
    hipLaunchKernelGGL(c4_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, as_stream(stream), src, bias, static_cast<unsigned char *>(dst),
                       static_cast<unsigned char *>(scales), R, K, src_ld, ones, weight);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" size_t cosa_c4_record_bytes(void) { return sizeof(C4Rec); }

/* c4 WEIGHT rows of several fp32 matrices (each [rows, K] contiguous, rows % 256 == 0, K % 256 == 0, optional bias [rows]) in one launch.
 * records: device array of { const float *src; const float *bias; void *dst; void *scales; int rows, K, row0, pad } */
extern "C" int cosa_c4_rows_batched(const void *records, int n_records, int total_rows, void *stream)
{
    COSA_REQUIRE(records && n_records > 0 && total_rows > 0, "cosa_c4_rows_batched: bad arguments");
    hipLaunchKernelGGL(c4_rows_batched_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, as_stream(stream), static_cast<const C4Rec *>(records),
                       n_records, total_rows);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_layernorm_c4(const float *x, const float *gamma, const float *beta, void *y_c4, void *y_scales, float *y_f32, int rows, int dim,
                                 float eps, void *stream)
{
    COSA_REQUIRE(x && gamma && beta && (y_c4 || y_f32) && (!y_c4 || y_scales) && rows > 0, "cosa_layernorm_c4: bad arguments");
    COSA_REQUIRE(dim == 768, "cosa_layernorm_c4: dim must be 768 (ViT-B)");
    hipLaunchKernelGGL(layernorm_c4_kernel<768>, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta,
                       static_cast<unsigned char *>(y_c4), static_cast<unsigned char *>(y_scales), y_f32, rows, eps);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

template <typename T>
static int split_rows_launch(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream)
{
    COSA_REQUIRE(src && dst && R > 0 && K > 0 && K % 4 == 0 && src_ld >= K && src_ld % 4 == 0, "cosa_split_rows: bad arguments (K %% 4 == 0)");
    hipLaunchKernelGGL(split_rows_kernel<T>, dim3((R + 3) / 4), dim3(256), 0, as_stream(stream), src, bias, static_cast<T *>(dst), R, K, src_ld, ones);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

template <typename T>
static int layernorm_split_launch(const float *x, const float *gamma, const float *beta, void *y_split, float *y_f32, int rows, int dim, float eps,
                                  void *stream)
{
    COSA_REQUIRE(x && gamma && beta && (y_split || y_f32) && rows > 0, "cosa_layernorm_split: bad arguments");
    COSA_REQUIRE(dim == 768, "cosa_layernorm_split: dim must be 768 (ViT-B)");
    hipLaunchKernelGGL((layernorm_split_kernel<768, T>), dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta,
                       static_cast<T *>(y_split), y_f32, rows, eps);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

template <typename T>
static int im2col_flip_split_launch(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream)
{
    COSA_REQUIRE(x && rows && B > 0 && C > 0 && H > 0 && W > 0 && P > 0 && cls_rows >= 0, "cosa_im2col_flip_split_tokens: bad arguments");
    COSA_REQUIRE(P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "cosa_im2col_flip_split_tokens: patch size must be a multiple of 8 and divide H and W (got P=%d H=%d W=%d)", P, H, W);
    COSA_REQUIRE((flips == 1 || flips == 2) && (C * P * P) % 64 == 0, "cosa_im2col_flip_split_tokens: flips 1 | 2, C*P*P %% 64 == 0");
    const size_t total = (size_t)flips * B * (H / P) * (W / P) * (C * P * P / 8);
    COSA_REQUIRE(total / 256 < 0x7fffffffull, "cosa_im2col_flip_split_tokens: too many elements for one launch");
    hipLaunchKernelGGL(im2col_flip_split_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, static_cast<T *>(rows), B, C, H, W, P,
                       flips, cls_rows);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

/* split rows (hi | lo | aug, bf16 halves; _f16: fp16 halves) of the im2col view of x [B,C,H,W] (flips = 1) or cat(x, x.flip(-1)) (flips = 2)
 * written INTO a token matrix [flips * B * (h*w + cls_rows), 2 C P P + 64]: image (f, b)'s patch rows start at row (f*B + b) * (h*w + cls_rows) +
 * cls_rows; the rows in between are not touched (vit.py:254-262 PatchEmbed as a GEMM over token rows; seg_helper.py:241-246 the flip pair) */
extern "C" int cosa_im2col_flip_split_tokens(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream)
{
    return im2col_flip_split_launch<bf16>(x, rows, B, C, H, W, P, flips, cls_rows, stream);
}
extern "C" int cosa_im2col_flip_split_tokens_f16(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream)
{
    return im2col_flip_split_launch<_Float16>(x, rows, B, C, H, W, P, flips, cls_rows, stream);
}

extern "C" int cosa_split_rows(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream)
{
    return split_rows_launch<bf16>(src, bias, dst, R, K, src_ld, ones, stream);
}
extern "C" int cosa_split_rows_f16(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream)
{
    return split_rows_launch<_Float16>(src, bias, dst, R, K, src_ld, ones, stream);
}

extern "C" int cosa_layernorm_split(const float *x, const float *gamma, const float *beta, void *y_split, float *y_f32, int rows, int dim,
                                    float eps, void *stream)
{
    return layernorm_split_launch<bf16>(x, gamma, beta, y_split, y_f32, rows, dim, eps, stream);
}
extern "C" int cosa_layernorm_split_f16(const float *x, const float *gamma, const float *beta, void *y_split, float *y_f32, int rows, int dim,
                                        float eps, void *stream)
{
    return layernorm_split_launch<_Float16>(x, gamma, beta, y_split, y_f32, rows, dim, eps, stream);
}

// ---- fp16c8 producers (c8.hpp; include/cosa_hip.h) ---------------------------------------------------------------------------------
extern "C" int cosa_c8_rows(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream)
{
    COSA_REQUIRE(src && dst && R > 0 && K > 0 && K % 128 == 0 && src_ld >= K && src_ld % 4 == 0, "cosa_c8_rows: bad arguments (K %% 128 == 0)");
    hipLaunchKernelGGL(c8_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, as_stream(stream), src, bias, static_cast<unsigned char *>(dst), R, K,
                       src_ld, ones);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_layernorm_c8(const float *x, const float *gamma, const float *beta, void *y_c8, float *y_f32, int rows, int dim,
                                 float eps, void *stream)
{
    COSA_REQUIRE(x && gamma && beta && (y_c8 || y_f32) && rows > 0, "cosa_layernorm_c8: bad arguments");
    COSA_REQUIRE(dim == 768, "cosa_layernorm_c8: dim must be 768 (ViT-B)");
    hipLaunchKernelGGL(layernorm_c8_kernel<768>, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, gamma, beta,
                       static_cast<unsigned char *>(y_c8), y_f32, rows, eps);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

/* c8 rows of several fp32 matrices (each [rows, K] contiguous, K % 128 == 0, optional bias [rows]) in one launch.  records: device array of
 * { const float *src; const float *bias; void *dst; int rows, K, row0, pad } (cosa_c8_record_bytes() each), row0 = running sum of rows */
extern "C" size_t cosa_c8_record_bytes(void) { return sizeof(C8Rec); }

extern "C" int cosa_c8_rows_batched(const void *records, int n_records, int total_rows, void *stream)
{
    COSA_REQUIRE(records && n_records > 0 && total_rows > 0, "cosa_c8_rows_batched: bad arguments");
    hipLaunchKernelGGL(c8_rows_batched_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, as_stream(stream), static_cast<const C8Rec *>(records),
                       n_records, total_rows);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
