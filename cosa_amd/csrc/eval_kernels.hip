// eval_kernels.hip -- evaluation path (SURVEY f-1): label maps at the ground truth's resolution and the confusion matrix.
//
//   evaluation_engine.py:96-126,198-200   F.interpolate(cam / seg, size=labels.shape) -> cam_to_label / seg_validation -> argmax
//   utils/seg_helper.py:515-546           cam_to_label
//   utils/evaluation.py:10-70             _fast_hist / scores / pseudo_scores
//
// HBM-bound byte work.  The reference materialises the resized [1,C,H,W] and [1,C+1,H,W] tensors (and clones of them) and takes
// three argmaxes over them; here one thread per output pixel samples the (S,S) maps (L2-resident: 448^2 x 4 B x (2C+1)) and writes
// three bytes.  Arithmetic spec R (this file is compiled with -ffp-contract=off):
//   scale = (float)in / (float)out;  src = max(fmaf(scale, dst + 0.5f, -0.5f), 0);  i0 = min((int)src, in-1);  l1 = src - i0
//   r0 = fma(p00, lx0, p01*lx1);  r1 = fma(p10, lx0, p11*lx1);  v = fma(r0, ly0, r1*ly1)
// which is how ATen's CPU kernel evaluates F.interpolate(bilinear, align_corners=False) at image sizes: the label maps are
// bit-identical to the reference's on the golden vectors (tests/golden/eval.npz).
#include "kernels.hpp"

namespace cosa {
namespace {

__device__ __forceinline__ void src_index_r(int dst, int in, float scale, int &i0, int &i1, float &l0, float &l1)
{
    float src = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int i = (int)src;
    i = i > in - 1 ? in - 1 : i;
    float lam = src - (float)i;
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    i0 = i;
    i1 = i < in - 1 ? i + 1 : i;
    l1 = lam;
    l0 = 1.0f - lam;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ pl, int w, int y0, int y1, int x0, int x1, float ly0, float ly1,
                                        float lx0, float lx1)
{
    const float r0 = __builtin_fmaf(pl[(size_t)y0 * w + x0], lx0, pl[(size_t)y0 * w + x1] * lx1);
    const float r1 = __builtin_fmaf(pl[(size_t)y1 * w + x0], lx0, pl[(size_t)y1 * w + x1] * lx1);
    return __builtin_fmaf(r0, ly0, r1 * ly1);
}

// cam [B,C,S,S], seg [B,C+1,S,S], cls [B,C] -> three uint8 maps [B,H,W]
__global__ __launch_bounds__(256) void eval_labels_kernel(const float *__restrict__ cam, const float *__restrict__ seg,
                                                         const float *__restrict__ cls, int C, int S, int H, int W, float sy, float sx,
                                                         float bkg_thre, uint8_t *__restrict__ lab_cam, uint8_t *__restrict__ lab_ps,
                                                         uint8_t *__restrict__ lab_vd)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int b = blockIdx.y;
    const int Y = pix / W, X = pix - Y * W;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index_r(Y, S, sy, y0, y1, ly0, ly1);
    src_index_r(X, S, sx, x0, x1, lx0, lx1);
    const size_t ss = (size_t)S * S;
    const float *cl = cls + (size_t)b * C;
    const size_t o = (size_t)b * H * W + pix;
    if (cam) {
        const float *cb = cam + (size_t)b * C * ss;
        float best = 0.0f;
        int bi = 0;
        for (int c = 0; c < C; c++) {
            const float l = cl[c];
            // an absent class contributes l * v = 0 exactly (CAMs are finite): skip its four taps
            const float v = l != 0.0f ? l * bilerp(cb + c * ss, S, y0, y1, x0, x1, ly0, ly1, lx0, lx1) : 0.0f;
            if (c == 0 || v > best) { best = v; bi = c; }
        }
        lab_cam[o] = best <= bkg_thre ? 0 : (uint8_t)(bi + 1);
    }
    if (seg) {
        const float *sb = seg + (size_t)b * (C + 1) * ss;
        float bp = 0.0f, bv = 0.0f;
        int ip = 0, iv = 0;
        for (int c = 0; c <= C; c++) {
            const float v = bilerp(sb + c * ss, S, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
            if (c == 0 || v > bp) { bp = v; ip = c; }
            const float vv = (c == 0 || cl[c - 1] != 0.0f) ? v : -1e5f;
            if (c == 0 || vv > bv) { bv = vv; iv = c; }
        }
        lab_ps[o] = (uint8_t)ip;
        lab_vd[o] = (uint8_t)iv;
    }
}

// cam_to_label on an already sized CAM [B,C,H,W]
__global__ __launch_bounds__(256) void cam_to_label_kernel(const float *__restrict__ cam, const float *__restrict__ cls, int C, int H, int W,
                                                          float bkg_thre, const int32_t *__restrict__ boxes, int ignore_mid, float high_thre,
                                                          float low_thre, long long ignore_index, long long *__restrict__ label,
                                                          float *__restrict__ valid_cam)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const float *cb = cam + (size_t)b * C * hw;
    float best = 0.0f;
    int bi = 0;
    for (int c = 0; c < C; c++) {
        float v = cb[c * hw + pix];
        if (cls) v = cls[(size_t)b * C + c] * v;
        if (valid_cam) valid_cam[((size_t)b * C + c) * hw + pix] = v;
        if (c == 0 || v > best) { best = v; bi = c; }
    }
    long long l = bi + 1;
    if (best <= bkg_thre) l = 0;
    if (boxes) {
        if (ignore_mid) {
            if (best <= high_thre) l = ignore_index;
            if (best <= low_thre) l = 0;
        }
        const int Y = pix / W, X = pix - Y * W;
        const int32_t *bx = boxes + 4 * b;
        if (!(Y >= bx[0] && Y < bx[1] && X >= bx[2] && X < bx[3])) l = ignore_index;
    }
    label[(size_t)b * hw + pix] = l;
}

// hist[nc*t + p] += 1 over pixels with t < nc.  Workgroup-private LDS counters (nc^2 <= 8192), flushed with one global atomic per
// non-zero bin: the 81 x 81 COCO matrix costs 26 KB of LDS.
constexpr int kHistLds = 8192;
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t *__restrict__ gt, const uint8_t *__restrict__ pred, size_t n, int nc,
                                                       int pseudo, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[kHistLds];
    const int bins = nc * nc;
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256 * 16;
    for (size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; base < n; base += stride) {
        if (base + 16 <= n && (((size_t)(gt + base) | (size_t)(pred + base)) & 15) == 0) {
            const uint4 g4 = *reinterpret_cast<const uint4 *>(gt + base), p4 = *reinterpret_cast<const uint4 *>(pred + base);
            const unsigned gw[4] = {g4.x, g4.y, g4.z, g4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int t = (gw[k >> 2] >> (8 * (k & 3))) & 255, p = (pw[k >> 2] >> (8 * (k & 3))) & 255;
                if (pseudo && p == 255) continue;
                if (t < nc && p < nc) atomicAdd(&h[nc * t + p], 1u);
            }
        } else {
            for (size_t i = base; i < n && i < base + 16; i++) {
                const int t = gt[i], p = pred[i];
                if (pseudo && p == 255) continue;
                if (t < nc && p < nc) atomicAdd(&h[nc * t + p], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

}  // namespace
}  // namespace cosa

using namespace cosa;

extern "C" int cosa_eval_labels(const float *cam, const float *seg, const float *cls_label, int B, int C, int S, int H, int W,
                                float bkg_thre, uint8_t *lab_cam, uint8_t *lab_ps, uint8_t *lab_vd, void *stream)
{
    COSA_REQUIRE(cls_label && (cam || seg) && B > 0 && C > 0 && C < 255 && S > 0 && H > 0 && W > 0, "cosa_eval_labels: bad arguments");
    COSA_REQUIRE((!cam || lab_cam) && (!seg || (lab_ps && lab_vd)), "cosa_eval_labels: missing output map");
    COSA_REQUIRE((size_t)H * W < 0x7fffffffull && B <= 65535, "cosa_eval_labels: map too large");
    const float sy = (float)S / (float)H, sx = (float)S / (float)W;
    hipLaunchKernelGGL(eval_labels_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, as_stream(stream), cam, seg, cls_label, C, S, H, W,
                       sy, sx, bkg_thre, lab_cam, lab_ps, lab_vd);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_cam_to_label(const float *cam, const float *cls_label, int B, int C, int H, int W, float bkg_thre,
                                 const int32_t *boxes, int ignore_mid, float high_thre, float low_thre, long long ignore_index,
                                 long long *label, float *valid_cam, void *stream)
{
    COSA_REQUIRE(cam && label && B > 0 && C > 0 && H > 0 && W > 0, "cosa_cam_to_label: bad arguments");
    COSA_REQUIRE((size_t)H * W < 0x7fffffffull && B <= 65535, "cosa_cam_to_label: map too large");
    hipLaunchKernelGGL(cam_to_label_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, as_stream(stream), cam, cls_label, C, H, W, bkg_thre,
                       boxes, ignore_mid, high_thre, low_thre, ignore_index, label, valid_cam);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}

extern "C" int cosa_confusion_hist(const uint8_t *gt, const uint8_t *pred, size_t n, int num_classes, int pseudo,
                                   unsigned long long *hist, void *stream)
{
    COSA_REQUIRE(gt && pred && hist && num_classes > 0, "cosa_confusion_hist: bad arguments");
    COSA_REQUIRE(num_classes * num_classes <= kHistLds, "cosa_confusion_hist: at most 90 classes (got %d)", num_classes);
    if (n == 0) return COSA_OK;
    size_t blocks = (n + 256 * 16 * 8 - 1) / (256 * 16 * 8);          // >= 8 chunks of 16 bytes per thread: amortises the flush
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), gt, pred, n, num_classes, pseudo, hist);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
