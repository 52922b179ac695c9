// label_kernels.hip -- CAM -> pseudo-label path (HBM-bound byte/index work), gfx950.
//
// Compiled with -ffp-contract=off: every float op below is one IEEE binary32 operation in the
// order written and FMAs appear only as __builtin_fmaf(), following DESIGN.md "Arithmetic spec".
// The label maps are therefore bit-identical to the CPU oracle.
//
// Reference: utils/seg_helper.py:232-275 (multi_scale_camseg tail), :547-551 (cam_validation),
//            :721-797 (cam2mask, _refine_cams), utils/torch_helper.py:354-367 (denormalize_img).
#include "kernels.hpp"

namespace cosa {
namespace {

// ---- spec E: deterministic expf --------------------------------------------------------
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

// ATen area_pixel_compute_source_index (align_corners=False) + guard, spec U
__device__ __forceinline__ void src_index(int dst, int in, int out, float scale, int &i0, int &i1, float &l0, float &l1)
{
    if (in == out) { i0 = dst; i1 = dst < in - 1 ? dst + 1 : dst; l0 = 1.0f; l1 = 0.0f; return; }
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int i = (int)src;
    if (i > in - 1) i = in - 1;
    i0 = i;
    i1 = i < in - 1 ? i + 1 : i;
    l1 = src - (float)i;
    l0 = 1.0f - l1;
}

__device__ __forceinline__ float bilerp(float p00, float p01, float p10, float p11, float lx0, float lx1, float ly0, float ly1)
{
    float r0 = __builtin_fmaf(p00, lx0, p01 * lx1);
    float r1 = __builtin_fmaf(p10, lx0, p11 * lx1);
    return __builtin_fmaf(r0, ly0, r1 * ly1);
}

// ---- denormalize_img ---------------------------------------------------------------------
__global__ void denorm_kernel(const float *__restrict__ img, float *__restrict__ out, int HW, size_t total)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int c = (int)((i / HW) % 3);
        float sd = c == 0 ? 58.395f : (c == 1 ? 57.12f : 57.375f);
        float mn = c == 0 ? 123.675f : (c == 1 ? 116.28f : 103.53f);
        float v = img[i] * sd;
        v = v + mn;
        int iv = (int)v;
        unsigned char u = (unsigned char)iv;
        out[i] = (float)u / 255.0f;
    }
}

// ---- block reductions (max) ---------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int NT>
__device__ __forceinline__ float block_max(float v, float *sh)
{
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; i++) r = fmaxf(r, sh[i]);
    return r;
}

// one 1024-thread block per (b,c) plane; float4 path when HW % 4 == 0
__global__ __launch_bounds__(1024) void minmax_norm_kernel(float *__restrict__ cam, int HW, const float *__restrict__ active)
{
    __shared__ float sh[16];
    if (active && active[blockIdx.x] == 0.0f) return;   // all-zero plane stays all-zero
    float *x = cam + (size_t)blockIdx.x * HW;
    float mneg = -INFINITY;
    if ((HW & 3) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int i = threadIdx.x; i < HW / 4; i += 1024) {
            float4 v = x4[i];
            mneg = fmaxf(mneg, fmaxf(fmaxf(-v.x, -v.y), fmaxf(-v.z, -v.w)));
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 1024) mneg = fmaxf(mneg, -x[i]);
    }
    mneg = block_max<1024>(mneg, sh);
    float mx = -INFINITY;
    if ((HW & 3) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int i = threadIdx.x; i < HW / 4; i += 1024) {
            float4 v = x4[i];
            mx = fmaxf(mx, fmaxf(fmaxf(v.x + mneg, v.y + mneg), fmaxf(v.z + mneg, v.w + mneg)));
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 1024) mx = fmaxf(mx, x[i] + mneg);
    }
    mx = block_max<1024>(mx, sh);
    const float den = mx + 1e-5f;
    if ((HW & 3) == 0) {
        float4 *x4 = reinterpret_cast<float4 *>(x);
        for (int i = threadIdx.x; i < HW / 4; i += 1024) {
            float4 v = x4[i];
            v.x = (v.x + mneg) / den; v.y = (v.y + mneg) / den; v.z = (v.z + mneg) / den; v.w = (v.w + mneg) / den;
            x4[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 1024) x[i] = (x[i] + mneg) / den;
    }
}

// The same normalisation with the plane spread over several workgroups (the one-block-per-plane kernel above keeps ~40 of 256 CUs busy when
// only the 2-3 classes of an image are live): pass 1 reduces min and max of the plane through order-preserving unsigned keys and integer
// atomics, pass 2 normalises.  max_x fl(x + mneg) = fl(max_x x + mneg) (rounding is monotonic), so the result is bit for bit the above.
__device__ __forceinline__ unsigned f2key(float f) { const unsigned u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

__global__ __launch_bounds__(256) void minmax_init_kernel(unsigned *__restrict__ kmin, unsigned *__restrict__ kmax, int BC)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < BC) { kmin[i] = 0xFFFFFFFFu; kmax[i] = 0u; }
}

__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float *__restrict__ cam, int HW, const float *__restrict__ active,
                                                           unsigned *__restrict__ kmin, unsigned *__restrict__ kmax)
{
    __shared__ float sh[8];
    const int plane = blockIdx.y;
    if (active && active[plane] == 0.0f) return;
    const float *x = cam + (size_t)plane * HW;
    const int per = (HW + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(HW, i0 + per);
    float mn = INFINITY, mx = -INFINITY;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mx = block_max<256>(mx, sh);
    __syncthreads();
    mn = -block_max<256>(-mn, sh);
    if (threadIdx.x == 0 && i0 < i1) {
        atomicMin(&kmin[plane], f2key(mn));
        atomicMax(&kmax[plane], f2key(mx));
    }
}

__global__ __launch_bounds__(256) void minmax_apply_kernel(float *__restrict__ cam, int HW, const float *__restrict__ active,
                                                          const unsigned *__restrict__ kmin, const unsigned *__restrict__ kmax)
{
    const int plane = blockIdx.y;
    if (active && active[plane] == 0.0f) return;
    float *x = cam + (size_t)plane * HW;
    const float mneg = -key2f(kmin[plane]);
    const float den = (key2f(kmax[plane]) + mneg) + 1e-5f;
    const int per = (HW + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(HW, i0 + per);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) x[i] = (x[i] + mneg) / den;
}

// ---- multi_scale_camseg tail: upsample + flip-merge (+relu) + accumulate ---------------------
// thread per output pixel; grid (ceil(S*S/256), C, B)
// grid (pixel blocks, images): the class planes are a loop INSIDE the workgroup.  Absent classes cost one uniform branch instead of a
// launched-and-retired workgroup each (b=16 x 80 COCO planes x 784 blocks were a million workgroups per call, 97 % of them empty), and the
// bilinear source indices / weights of a pixel are computed once for all its planes.
__global__ __launch_bounds__(256) void flip_merge_upsample_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                 int B, int C, int h, int w, int S, float sy, float sx,
                                                                 int mode, int accumulate, const float *__restrict__ active,
                                                                 const float *__restrict__ prev_active)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    // the live planes (and those to clear) of this image as bit masks, built once per wave by ballot: the plane loop then visits only them
    // (80 COCO planes with ~3 live: the per-plane test of the activity flag was most of the kernel's time)
    unsigned long long live[2] = {0ull, 0ull}, clr[2] = {0ull, 0ull};
    const bool masked = active != nullptr && C <= 128;
    if (masked) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int wd = 0; wd < 2; wd++) {
            const int c = wd * 64 + lane;
            const float a = c < C ? active[b * C + c] : 0.0f;
            const float pv = (c < C && prev_active) ? prev_active[b * C + c] : 1.0f;
            live[wd] = __ballot(c < C && a != 0.0f);
            clr[wd] = __ballot(c < C && a == 0.0f && pv != 0.0f);
        }
    }
    if (pix >= S * S) return;
    const int Y = pix / S, X = pix - Y * S;
    int y0, y1, x0, x1, fx0, fx1;
    float ly0, ly1, lx0, lx1, flx0, flx1;
    src_index(Y, h, S, sy, y0, y1, ly0, ly1);
    src_index(X, w, S, sx, x0, x1, lx0, lx1);
    src_index(S - 1 - X, w, S, sx, fx0, fx1, flx0, flx1);
    if (masked && !accumulate) {
        // class absent from the image: cam_validation zeroes it anyway.  prev_active: dst is a buffer that this routine filled last time
        // under that activity map, so an absent plane is already zero unless it was live then
#pragma unroll
        for (int wd = 0; wd < 2; wd++)
            for (unsigned long long mk = clr[wd]; mk; mk &= mk - 1) {
                const int c = wd * 64 + __builtin_ctzll(mk);
                dst[((size_t)b * C + c) * S * S + pix] = 0.0f;
            }
    }
    for (int it = 0, c = -1; it < C; it++) {
        if (masked) {                                   // next live plane
            const int wd = live[0] ? 0 : 1;
            if (!live[wd]) break;
            c = wd * 64 + __builtin_ctzll(live[wd]);
            live[wd] &= live[wd] - 1;
        } else {
            c = it;
            if (active && active[b * C + c] == 0.0f) {
                if (!accumulate && (!prev_active || prev_active[b * C + c] != 0.0f)) dst[((size_t)b * C + c) * S * S + pix] = 0.0f;
                continue;
            }
        }
        const size_t o = ((size_t)b * C + c) * S * S + pix;
        const float *p = src + ((size_t)b * C + c) * h * w;
        const float *q = src + ((size_t)(b + B) * C + c) * h * w;
        const float u1 = bilerp(p[y0 * w + x0], p[y0 * w + x1], p[y1 * w + x0], p[y1 * w + x1], lx0, lx1, ly0, ly1);
        const float u2 = bilerp(q[y0 * w + fx0], q[y0 * w + fx1], q[y1 * w + fx0], q[y1 * w + fx1], flx0, flx1, ly0, ly1);
        float v;
        if (mode == 0) { v = fmaxf(u1, u2); v = v > 0.0f ? v : 0.0f; }
        else v = u1 + u2;
        dst[o] = accumulate ? dst[o] + v : v;
    }
}

// ---- active channel list -------------------------------------------------------------------
__global__ void active_kernel(const float *__restrict__ labels, int *__restrict__ act, int *__restrict__ kcount, int B, int C)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int *a = act + (size_t)b * (C + 1);
    int K = 0;
    a[K++] = 0;
    for (int c = 0; c < C; c++)
        if (labels[(size_t)b * C + c] != 0.0f) a[K++] = c + 1;
    kcount[b] = K;
    kcount[B + b] = 2 * K;   // hi+lo stacked plane count, used by the PAR step
}

// ---- spec L: threshold plane + /2 bilinear + active softmax (hi and lo in one pass) ---------------
// P layout: [B][H][Kmax][s*s], Kmax = C+1; a CAM set owns halves h0 (hi planes) and h0+1 (lo planes); H = 2 x CAM sets
template <int DS>
__device__ __forceinline__ float lowres_value(const float *__restrict__ pl, float lab, int S, int y, int x)
{
    if (DS == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(pl + (size_t)(2 * y) * S + 2 * x);
        const float2 u = *reinterpret_cast<const float2 *>(pl + (size_t)(2 * y + 1) * S + 2 * x);
        float a00 = t.x * lab, a01 = t.y * lab, a10 = u.x * lab, a11 = u.y * lab;
        float r0 = a00 * 0.5f + a01 * 0.5f;
        float r1 = a10 * 0.5f + a11 * 0.5f;
        return r0 * 0.5f + r1 * 0.5f;
    } else {
        return pl[(size_t)y * S + x] * lab;
    }
}

template <int DS>
__global__ __launch_bounds__(256) void lowres_softmax_kernel(const float *__restrict__ cams, const float *__restrict__ labels,
                                                            const int *__restrict__ act, const int *__restrict__ kcount,
                                                            float *__restrict__ P, int C, int S, int s, float thr_hi, float thr_lo, int fold,
                                                            int H, int h0, const float *__restrict__ thr_dev)
{
    if (thr_dev) { thr_hi = thr_dev[h0]; thr_lo = thr_dev[h0 + 1]; }       // device-resident thresholds of this CAM set
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= s * s) return;
    const int b = blockIdx.y;
    const int y = pix / s, x = pix - y * s;
    const int K = kcount[b];
    const int Kmax = C + 1;
    const int *a = act + (size_t)b * Kmax;
    const float *cam = cams + (size_t)b * C * S * S;
    const float *lab = labels + (size_t)b * C;
    const size_t ss = (size_t)s * s;
    float *Phi = P + ((size_t)b * H + h0 + 0) * Kmax * ss + pix;
    float *Plo = P + ((size_t)b * H + h0 + 1) * Kmax * ss + pix;

    float mc = -INFINITY;
    for (int k = 1; k < K; k++) {
        const int c = a[k] - 1;
        float v = lowres_value<DS>(cam + (size_t)c * S * S, fold ? lab[c] : 1.0f, S, y, x);
        mc = v > mc ? v : mc;
    }
    const float mhi = mc > thr_hi ? mc : thr_hi;
    const float mlo = mc > thr_lo ? mc : thr_lo;
    float ehi0 = cosa_expf(thr_hi - mhi), elo0 = cosa_expf(thr_lo - mlo);
    float shi = 0.0f + ehi0, slo = 0.0f + elo0;
    for (int k = 1; k < K; k++) {
        const int c = a[k] - 1;
        float v = lowres_value<DS>(cam + (size_t)c * S * S, fold ? lab[c] : 1.0f, S, y, x);
        shi = shi + cosa_expf(v - mhi);
        slo = slo + cosa_expf(v - mlo);
    }
    Phi[0] = ehi0 / shi;
    Plo[0] = elo0 / slo;
    for (int k = 1; k < K; k++) {
        const int c = a[k] - 1;
        float v = lowres_value<DS>(cam + (size_t)c * S * S, fold ? lab[c] : 1.0f, S, y, x);
        Phi[(size_t)k * ss] = cosa_expf(v - mhi) / shi;
        Plo[(size_t)k * ss] = cosa_expf(v - mlo) / slo;
    }
}

// ---- /2 bilinear of the de-normalised image for the PAR hook (utils/seg_helper.py:735-737) ------
__global__ __launch_bounds__(256) void halve_image_kernel(const float *__restrict__ img, float *__restrict__ out, int planes, int S)
{
    const int s = S / 2;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)planes * s * s) return;
    const int pl = (int)(i / ((size_t)s * s));
    const int pix = (int)(i - (size_t)pl * s * s);
    const int y = pix / s, x = pix - y * s;
    const float *p = img + (size_t)pl * S * S;
    const float2 t = *reinterpret_cast<const float2 *>(p + (size_t)(2 * y) * S + 2 * x);
    const float2 u = *reinterpret_cast<const float2 *>(p + (size_t)(2 * y + 1) * S + 2 * x);
    float r0 = t.x * 0.5f + t.y * 0.5f;
    float r1 = u.x * 0.5f + u.y * 0.5f;
    out[i] = r0 * 0.5f + r1 * 0.5f;
}

// ---- spec U: upsample + first-max argmax + key gather + box + hi/lo merge ------------------------
__device__ __forceinline__ int argmax_up(const float *__restrict__ Pb, int K, size_t ss, int s,
                                         int y0, int y1, int x0, int x1, float ly0, float ly1, float lx0, float lx1)
{
    float best = 0.0f;
    int bi = 0;
    for (int k = 0; k < K; k++) {
        const float *pl = Pb + (size_t)k * ss;
        float v = bilerp(pl[y0 * s + x0], pl[y0 * s + x1], pl[y1 * s + x0], pl[y1 * s + x1], lx0, lx1, ly0, ly1);
        if (k == 0 || v > best) { best = v; bi = k; }
    }
    return bi;
}

__global__ __launch_bounds__(256) void upsample_argmax_merge_kernel(const float *__restrict__ P, const int *__restrict__ act,
                                                                   const int *__restrict__ kcount, const int32_t *__restrict__ boxes,
                                                                   float *__restrict__ mask, int C, int S, int s, float scale,
                                                                   float ignore_index, int H, int h0)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= S * S) return;
    const int b = blockIdx.y;
    const int Y = pix / S, X = pix - Y * S;
    const int32_t *bx = boxes + b * 4;
    float r = ignore_index;
    if (Y >= bx[0] && Y < bx[1] && X >= bx[2] && X < bx[3]) {
        const int K = kcount[b];
        const int Kmax = C + 1;
        const size_t ss = (size_t)s * s;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(Y, s, S, scale, y0, y1, ly0, ly1);
        src_index(X, s, S, scale, x0, x1, lx0, lx1);
        const int *a = act + (size_t)b * Kmax;
        const int khi = argmax_up(P + ((size_t)b * H + h0 + 0) * Kmax * ss, K, ss, s, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
        const int klo = argmax_up(P + ((size_t)b * H + h0 + 1) * Kmax * ss, K, ss, s, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
        const float hi = (float)a[khi], lo = (float)a[klo];
        r = hi;
        if (hi == 0.0f) r = ignore_index;
        if (hi + lo == 0.0f) r = 0.0f;
    }
    mask[(size_t)b * S * S + pix] = r;
}

}  // namespace
}  // namespace cosa

using namespace cosa;

namespace {
// F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=False) on NCHW fp32 planes -- the teacher's input rescale (utils/seg_helper.py:247-250).
// ATen's formula (area_pixel_compute_source_index: src = scale (dst + 0.5) - 0.5 clamped at 0, scale = in / out; the blend
// h0 (w0 a + w1 b) + h1 (w0 c + w1 d)).  The source index is ONE fused multiply-add, as in ATen's own CPU and GPU builds (both are compiled
// with contraction; evaluated as a product and a difference the index is off by up to an ulp of the image size and the result by 3e-5 of
// the image range); written as an explicit fmaf because this translation unit is built with -ffp-contract=off.  Against the CPU operator:
// within 1 ulp of the image range (tests/test_label_gpu.py::test_resize_bilinear_is_atens_formula).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ src, float *__restrict__ dst, int planes, int H, int W,
                                                             int OH, int OW, float rh, float rw)
{
    const size_t n = (size_t)planes * OH * OW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const size_t pl = i / ((size_t)OW * OH);
        float h1r = __builtin_fmaf(rh, (float)oy + 0.5f, -0.5f);
        h1r = h1r < 0.f ? 0.f : h1r;
        float w1r = __builtin_fmaf(rw, (float)ox + 0.5f, -0.5f);
        w1r = w1r < 0.f ? 0.f : w1r;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.0f - h1l, w1l = w1r - (float)w1, w0l = 1.0f - w1l;
        const float *p = src + pl * (size_t)H * W + (size_t)h1 * W + w1;
        const float top = __builtin_fmaf(w0l, p[0], w1l * p[w1p]);
        const float bot = __builtin_fmaf(w0l, p[(size_t)h1p * W], w1l * p[(size_t)h1p * W + w1p]);
        dst[i] = __builtin_fmaf(h0l, top, h1l * bot);
    }
}
}  // namespace

extern "C" int cosa_resize_bilinear(const float *src, float *dst, int planes, int H, int W, int OH, int OW, void *stream)
{
    COSA_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "cosa_resize_bilinear: bad arguments");
    const size_t n = (size_t)planes * OH * OW;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, planes, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_denormalize_img(const float *img, float *out, int B, int H, int W, void *stream)
{
    COSA_REQUIRE(img && out && B > 0 && H > 0 && W > 0, "cosa_denormalize_img: bad arguments");
    const size_t total = (size_t)B * 3 * H * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(denorm_kernel, dim3(grid), dim3(256), 0, as_stream(stream), img, out, H * W, total);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_cam_minmax_norm(float *cam, int BC, int HW, const float *active, void *stream)
{
    COSA_REQUIRE(cam && BC > 0 && HW > 0, "cosa_cam_minmax_norm: bad arguments");
    hipLaunchKernelGGL(minmax_norm_kernel, dim3(BC), dim3(1024), 0, as_stream(stream), cam, HW, active);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// the same, a plane spread over several workgroups; workspace: 2 * BC unsigned (min keys, then max keys), written by this call
extern "C" int cosa_cam_minmax_norm_ws(float *cam, int BC, int HW, const float *active, void *workspace, void *stream)
{
    COSA_REQUIRE(cam && workspace && BC > 0 && HW > 0 && BC <= 65535, "cosa_cam_minmax_norm_ws: bad arguments");
    hipStream_t st = as_stream(stream);
    unsigned *kmin = static_cast<unsigned *>(workspace), *kmax = kmin + BC;
    // The keys are initialised by a KERNEL, not by hipMemsetAsync: this call sits inside the teacher's captured hipGraph, and memset nodes of
    // a captured graph are not ordered with the kernel nodes around them from the second replay on (ROCm 7.0: the second normalisation of a
    // pass -- same workspace block, reused by the allocator -- saw its keys reset while the first one's apply kernel was still to read
    // them, or its own reduce ran before the reset: NaN planes in every replay but the first.  Round 6; tests/test_label_gpu.py).
    hipLaunchKernelGGL(minmax_init_kernel, dim3((BC + 255) / 256), dim3(256), 0, st, kmin, kmax, BC);
    COSA_LAUNCH_CHECK();
    int chunks = (HW + 8191) / 8192;                 // >= 32 elements per thread
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    hipLaunchKernelGGL(minmax_reduce_kernel, dim3(chunks, BC), dim3(256), 0, st, cam, HW, active, kmin, kmax);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(minmax_apply_kernel, dim3(chunks, BC), dim3(256), 0, st, cam, HW, active, kmin, kmax);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_cam_flip_merge_upsample(const float *src, float *dst, int B, int C, int h, int w, int S,
                                            int mode, int accumulate, const float *active, void *stream)
{
    COSA_REQUIRE(src && dst && B > 0 && C > 0 && h > 0 && w > 0 && S > 0, "cosa_cam_flip_merge_upsample: bad arguments");
    COSA_REQUIRE(mode == 0 || mode == 1, "cosa_cam_flip_merge_upsample: mode must be 0 or 1");
    COSA_REQUIRE(C <= 65535 && B <= 65535, "cosa_cam_flip_merge_upsample: grid too large");
    const float sy = (float)h / (float)S, sx = (float)w / (float)S;
    dim3 grid((S * S + 255) / 256, B);
    hipLaunchKernelGGL(flip_merge_upsample_kernel, grid, dim3(256), 0, as_stream(stream), src, dst, B, C, h, w, S, sy, sx,
                       mode, accumulate, active, static_cast<const float *>(nullptr));
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// the same for a destination that the previous call of this routine filled with the activity map prev_active ([B*C]): planes absent then
// and now are not written at all (they are still zero)
extern "C" int cosa_cam_flip_merge_upsample_reuse(const float *src, float *dst, int B, int C, int h, int w, int S,
                                                  int mode, int accumulate, const float *active, const float *prev_active, void *stream)
{
    COSA_REQUIRE(src && dst && active && prev_active && B > 0 && C > 0 && h > 0 && w > 0 && S > 0, "cosa_cam_flip_merge_upsample_reuse: bad arguments");
    COSA_REQUIRE(mode == 0 || mode == 1, "cosa_cam_flip_merge_upsample_reuse: mode must be 0 or 1");
    COSA_REQUIRE(C <= 65535 && B <= 65535, "cosa_cam_flip_merge_upsample_reuse: grid too large");
    const float sy = (float)h / (float)S, sx = (float)w / (float)S;
    dim3 grid((S * S + 255) / 256, B);
    hipLaunchKernelGGL(flip_merge_upsample_kernel, grid, dim3(256), 0, as_stream(stream), src, dst, B, C, h, w, S, sy, sx,
                       mode, accumulate, active, prev_active);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

// workspace layout of cam2mask: act[B*(C+1)] | kcount[2B] | P[B][2G][C+1][s*s] | P2 (same) | img_lo[B*3*s*s] | aff[B][NN][s*s]
extern "C" size_t cosa_cam2mask_multi_workspace_bytes(int G, int B, int C, int S, int downscale, int n_dil)
{
    const int s = downscale ? S / downscale : S;
    const size_t ss = (size_t)s * s;
    size_t bytes = 0;
    bytes += align_up((size_t)B * (C + 1) * sizeof(int), 256);
    bytes += align_up((size_t)2 * B * sizeof(int), 256);
    bytes += align_up((size_t)B * 2 * G * (C + 1) * ss * sizeof(float), 256);
    if (n_dil > 0) {
        bytes += align_up((size_t)B * 2 * G * (C + 1) * ss * sizeof(float), 256);
        bytes += align_up((size_t)B * 3 * ss * sizeof(float), 256);
        bytes += align_up((size_t)B * n_dil * 8 * ss * sizeof(float), 256);
    }
    return bytes;
}

extern "C" size_t cosa_cam2mask_workspace_bytes(int B, int C, int S, int downscale, int n_dil)
{
    return cosa_cam2mask_multi_workspace_bytes(1, B, C, S, downscale, n_dil);
}

// G CAM sets of the SAME images (main and aux CAMs in training) become G label maps in one pass: the class bookkeeping,
// the half-resolution image and -- the point -- PAR's affinity tensor are built once, and every propagation step streams
// the affinities once for the hi and lo stacks of all sets (2G·K live planes per image).  Per set the arithmetic is that
// of cosa_cam2mask, so the outputs are bit-identical to G separate calls.
extern "C" int cosa_cam2mask_multi(const float *images, const int32_t *boxes, const float *const *cams, const float *labels,
                                   float *const *masks, const float *thr_hi, const float *thr_lo, const float *thr_dev, int G,
                                   int B, int C, int S, int downscale, int fold_validation, const int *dilations, int n_dil, int par_iters,
                                   float ignore_index, void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(boxes && cams && labels && masks && thr_hi && thr_lo && workspace, "cosa_cam2mask: null pointer");
    COSA_REQUIRE(G > 0 && G <= 8, "cosa_cam2mask: 1..8 CAM sets");
    for (int g = 0; g < G; g++) COSA_REQUIRE(cams[g] && masks[g], "cosa_cam2mask: null CAM set / mask pointer");
    COSA_REQUIRE(B > 0 && C > 0 && S > 0 && B <= 65535, "cosa_cam2mask: bad shape");
    COSA_REQUIRE(downscale == 0 || downscale == 2, "cosa_cam2mask: downscale must be 0 or 2");
    COSA_REQUIRE(!downscale || (S % 2) == 0, "cosa_cam2mask: S must be even when downscale=2");
    const bool use_par = par_iters > 0;
    if (!use_par) n_dil = 0;
    COSA_REQUIRE(!use_par || (images && dilations && n_dil > 0 && n_dil <= kMaxDil), "cosa_cam2mask: PAR needs images and 1..8 dilations");
    if (workspace_bytes < cosa_cam2mask_multi_workspace_bytes(G, B, C, S, downscale, n_dil)) {
        set_error("cosa_cam2mask: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    const int s = downscale ? S / downscale : S;
    const size_t ss = (size_t)s * s;
    const int Kmax = C + 1;
    const int H = 2 * G;
    Carver cv(workspace);
    int *act = cv.take<int>((size_t)B * Kmax);
    int *kcount = cv.take<int>((size_t)2 * B);
    float *P = cv.take<float>((size_t)B * H * Kmax * ss);

    hipLaunchKernelGGL(active_kernel, dim3((B + 63) / 64), dim3(64), 0, st, labels, act, kcount, B, C);
    COSA_LAUNCH_CHECK();
    dim3 g1((unsigned)((ss + 255) / 256), B);
    for (int g = 0; g < G; g++) {
        if (downscale == 2)
            hipLaunchKernelGGL(lowres_softmax_kernel<2>, g1, dim3(256), 0, st, cams[g], labels, act, kcount, P, C, S, s, thr_hi[g],
                               thr_lo[g], fold_validation, H, 2 * g, thr_dev);
        else
            hipLaunchKernelGGL(lowres_softmax_kernel<0>, g1, dim3(256), 0, st, cams[g], labels, act, kcount, P, C, S, s, thr_hi[g],
                               thr_lo[g], fold_validation, H, 2 * g, thr_dev);
        COSA_LAUNCH_CHECK();
    }

    const float *Pfinal = P;
    if (use_par) {
        float *P2 = cv.take<float>((size_t)B * H * Kmax * ss);
        float *img_lo = cv.take<float>((size_t)B * 3 * ss);
        float *aff = cv.take<float>((size_t)B * n_dil * 8 * ss);
        ParPlan plan;
        int rc = par_make_plan(dilations, n_dil, &plan);
        if (rc) return rc;
        const float *im = images;
        if (downscale == 2) {
            const size_t tot = (size_t)B * 3 * ss;
            hipLaunchKernelGGL(halve_image_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, images, img_lo, B * 3, S);
            COSA_LAUNCH_CHECK();
            im = img_lo;
        }
        rc = par_launch_affinity(im, aff, B, s, s, plan, st);
        if (rc) return rc;
        float *src = P, *dst = P2;
        for (int it = 0; it < par_iters; it++) {
            // the H stacks of an image are contiguous: H*Kmax planes, of which the first K of each stack are live
            rc = par_launch_step(aff, src, dst, B, H * Kmax, kcount, H, (size_t)H * Kmax * ss, s, s, plan, st);
            if (rc) return rc;
            float *t = src; src = dst; dst = t;
        }
        Pfinal = src;
    }
    dim3 g3((S * S + 255) / 256, B);
    const float scale = (float)s / (float)S;
    for (int g = 0; g < G; g++) {
        hipLaunchKernelGGL(upsample_argmax_merge_kernel, g3, dim3(256), 0, st, Pfinal, act, kcount, boxes, masks[g], C, S, s, scale,
                           ignore_index, H, 2 * g);
        COSA_LAUNCH_CHECK();
    }
    return COSA_OK;
}

extern "C" int cosa_cam2mask(const float *images, const int32_t *boxes, const float *cams, const float *labels,
                             float *mask, int B, int C, int S, float thr_hi, float thr_lo, int downscale,
                             int fold_validation, const int *dilations, int n_dil, int par_iters, float ignore_index,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    COSA_REQUIRE(cams && mask, "cosa_cam2mask: null pointer");
    return cosa_cam2mask_multi(images, boxes, &cams, labels, &mask, &thr_hi, &thr_lo, nullptr, 1, B, C, S, downscale, fold_validation,
                               dilations, n_dil, par_iters, ignore_index, workspace, workspace_bytes, stream);
}
