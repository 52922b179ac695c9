// aug_kernels.hip -- the training input pipeline on the device (SURVEY f-2).
//
//   dataloaders/voc.py:262-275        __transforms: random_scaling -> random_fliplr -> random_crop -> GaussianBlur(p=.5) ->
//                                     weak = normalize(img); strong = normalize(OneOf(9 ops)(img))
//   dataloaders/transforms.py:10-28,52-77,104-120,150-202;  dataloaders/randaug.py:58-130
//
// The reference runs this per image on one CPU worker through Pillow (63 img/s per core measured; a GPU consumes 330).  The host
// keeps what must stay there -- JPEG decode and the random draws, in the reference's order -- and ships the decoded uint8
// image plus a parameter record; everything else happens here, and because the reference's arithmetic is integer / fixed
// point (Pillow), it is reproduced BIT FOR BIT (this file is compiled without contraction and without fast-math):
//
//   resize      Pillow BILINEAR: horizontal then vertical pass, 8-bit intermediate; per output index the triangle weights
//               over [centre - support, centre + support) are normalised in double and quantised to 22-bit fixed point.
//               Only what the crop window needs is resampled: flip, padding and cropping are index arithmetic.
//   blur        Pillow GaussianBlur = three horizontal + three vertical extended-box passes, each rounded to 8 bits:
//               (sum(2r+1 taps) * ww + (two outer taps) * fw + 2^23) >> 24 with edge replication.
//   strong ops  histogram look-up tables (autocontrast, equalize, solarize, posterize) or float32 blends with a degenerate
//               image (Color: ITU-R 601 luma, Contrast: its rounded mean, Brightness: black, Sharpness: 3x3 SMOOTH).
//   normalise   torchvision ToTensor + Normalize: (x / 255 - mean) / std in float32, divisions as divisions.
//
// All of it is HBM/L2-bound byte work on ~10 MB per batch: seven small launches, ~0.2 ms per 16 images.
#include "kernels.hpp"

#include <cstdint>

namespace cosa {
namespace {

constexpr int kAugMaxTaps = 16;

struct AugImage {                 // one record per image (device array; filled by the host, see cosa_amd/dataloaders/augment.py)
    long long raw_off;            // byte offset of the image in the packed uint8 HWC buffer
    int h, w, new_h, new_w;       // decoded size, size after random_scaling
    int flip;                     // random_fliplr
    int H_pad, W_pad, H_start, W_start;   // random_crop: position in the padded canvas, crop window
    int box[4];                   // img_box: rows [box0, box1), columns [box2, box3) of the crop that hold image
    int src_y0, src_rows;         // source rows the vertical pass reads (a superset is fine)
    int blur, br, bww, bfw;       // GaussianBlur: on/off, integer box radius, centre and outer weights (24-bit fixed point)
    int op, magnitude;            // strong op 0..8, magnitude 1..9
    float alpha;                  // ImageEnhance factor for the blend ops
    int pad_;
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for ONE output index (BILINEAR: support 1).
__device__ __forceinline__ void resample_taps(int xx, int in_size, int out_size, int &xmin, int &cnt, int (&kk)[kAugMaxTaps])
{
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fs;
    const double ss = 1.0 / fs;
    const double center = (xx + 0.5) * scale;
    xmin = (int)(center - support + 0.5);
    xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(center + support + 0.5);
    xmax = xmax > in_size ? in_size : xmax;
    cnt = xmax - xmin;
    cnt = cnt > kAugMaxTaps ? kAugMaxTaps : cnt;                    // the host refuses scales that would need more taps
    double w[kAugMaxTaps];
    double ww = 0.0;
    for (int x = 0; x < cnt; x++) {
        double a = (x + xmin - center + 0.5) * ss;
        a = a < 0.0 ? -a : a;
        const double v = a < 1.0 ? 1.0 - a : 0.0;
        w[x] = v;
        ww += v;
    }
    for (int x = 0; x < cnt; x++) {
        const double v = ww != 0.0 ? w[x] / ww : w[x];
        kk[x] = (int)(0.5 + v * 4194304.0);                         // 1 << 22
    }
}

__device__ __forceinline__ uint8_t clip8_fixed(long long acc)
{
    const long long v = acc >> 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass for the crop columns that hold image, over the source rows the vertical pass will read:
//   T[b][j][X] = resample_x(src row src_y0 + j) at scaled column xs = X + W_start - W_pad (mirrored when flipped)
__global__ __launch_bounds__(256) void aug_hpass_kernel(const uint8_t *__restrict__ raw, const AugImage *__restrict__ imgs,
                                                       uint8_t *__restrict__ T, int S, int max_rows)
{
    const AugImage im = imgs[blockIdx.z];
    const int X = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (j >= im.src_rows || X < im.box[2] || X >= im.box[3]) return;
    int xs = X + im.W_start - im.W_pad;
    if (im.flip) xs = im.new_w - 1 - xs;
    const uint8_t *row = raw + im.raw_off + (size_t)(im.src_y0 + j) * im.w * 3;
    uint8_t *out = T + (((size_t)blockIdx.z * max_rows + j) * S + X) * 3;
    if (im.new_w == im.w) {                                         // Pillow skips the pass
        out[0] = row[xs * 3]; out[1] = row[xs * 3 + 1]; out[2] = row[xs * 3 + 2];
        return;
    }
    int xmin, cnt, kk[kAugMaxTaps];
    resample_taps(xs, im.w, im.new_w, xmin, cnt, kk);
    long long a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int x = 0; x < cnt; x++) {
        const uint8_t *p = row + (size_t)(xmin + x) * 3;
        a0 += (long long)p[0] * kk[x];
        a1 += (long long)p[1] * kk[x];
        a2 += (long long)p[2] * kk[x];
    }
    out[0] = clip8_fixed(a0); out[1] = clip8_fixed(a1); out[2] = clip8_fixed(a2);
}

// vertical pass + padding: crop[b][Y][X] (HWC uint8); outside img_box the canvas is 0 (mean_rgb = [0,0,0])
__global__ __launch_bounds__(256) void aug_vpass_kernel(const AugImage *__restrict__ imgs, const uint8_t *__restrict__ T,
                                                       uint8_t *__restrict__ crop, int S, int max_rows)
{
    const AugImage im = imgs[blockIdx.z];
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= S) return;
    uint8_t *out = crop + (((size_t)blockIdx.z * S + Y) * S + X) * 3;
    if (Y < im.box[0] || Y >= im.box[1] || X < im.box[2] || X >= im.box[3]) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    const int ys = Y + im.H_start - im.H_pad;
    const uint8_t *col = T + ((size_t)blockIdx.z * max_rows * S + X) * 3;
    if (im.new_h == im.h) {
        const uint8_t *p = col + (size_t)(ys - im.src_y0) * S * 3;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        return;
    }
    int ymin, cnt, kk[kAugMaxTaps];
    resample_taps(ys, im.h, im.new_h, ymin, cnt, kk);
    long long a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int y = 0; y < cnt; y++) {
        int j = ymin + y - im.src_y0;
        j = j < 0 ? 0 : (j >= im.src_rows ? im.src_rows - 1 : j);  // never taken when the host range is right; keeps reads in bounds
        const uint8_t *p = col + (size_t)j * S * 3;
        a0 += (long long)p[0] * kk[y];
        a1 += (long long)p[1] * kk[y];
        a2 += (long long)p[2] * kk[y];
    }
    out[0] = clip8_fixed(a0); out[1] = clip8_fixed(a1); out[2] = clip8_fixed(a2);
}

// one extended-box pass along x (axis = 1) or y (axis = 0); images without blur are copied
__global__ __launch_bounds__(256) void aug_box_pass_kernel(const AugImage *__restrict__ imgs, const uint8_t *__restrict__ src,
                                                          uint8_t *__restrict__ dst, int S, int axis)
{
    const AugImage im = imgs[blockIdx.z];
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= S) return;
    const uint8_t *base = src + (size_t)blockIdx.z * S * S * 3;
    const size_t o = ((size_t)blockIdx.z * S * S + (size_t)Y * S + X) * 3;
    if (!im.blur) {
        dst[o] = src[o]; dst[o + 1] = src[o + 1]; dst[o + 2] = src[o + 2];
        return;
    }
    const int pos = axis ? X : Y, stride = axis ? 3 : S * 3;
    const uint8_t *line = base + (axis ? (size_t)Y * S * 3 : (size_t)X * 3);
    const int r = im.br, last = S - 1;
    unsigned acc0 = 0, acc1 = 0, acc2 = 0;
    for (int d = -r; d <= r; d++) {
        int q = pos + d;
        q = q < 0 ? 0 : (q > last ? last : q);
        const uint8_t *p = line + (size_t)q * stride;
        acc0 += p[0]; acc1 += p[1]; acc2 += p[2];
    }
    int ql = pos - r - 1, qr = pos + r + 1;
    ql = ql < 0 ? 0 : ql;
    qr = qr > last ? last : qr;
    const uint8_t *pl = line + (size_t)ql * stride, *pr = line + (size_t)qr * stride;
    const unsigned ww = (unsigned)im.bww, fw = (unsigned)im.bfw;
    dst[o] = (uint8_t)((acc0 * ww + ((unsigned)pl[0] + pr[0]) * fw + (1u << 23)) >> 24);
    dst[o + 1] = (uint8_t)((acc1 * ww + ((unsigned)pl[1] + pr[1]) * fw + (1u << 23)) >> 24);
    dst[o + 2] = (uint8_t)((acc2 * ww + ((unsigned)pl[2] + pr[2]) * fw + (1u << 23)) >> 24);
}

__device__ __forceinline__ int luma601(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// histograms of R, G, B and luma of the weak image: hist[b][4][256]
__global__ __launch_bounds__(256) void aug_hist_kernel(const uint8_t *__restrict__ weak, unsigned *__restrict__ hist, int S)
{
    __shared__ unsigned h[4 * 256];
    for (int i = threadIdx.x; i < 1024; i += 256) h[i] = 0;
    __syncthreads();
    const size_t n = (size_t)S * S;
    const uint8_t *img = weak + (size_t)blockIdx.y * n * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
        atomicAdd(&h[r], 1u);
        atomicAdd(&h[256 + g], 1u);
        atomicAdd(&h[512 + b], 1u);
        atomicAdd(&h[768 + luma601(r, g, b)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256)
        if (h[i]) atomicAdd(&hist[(size_t)blockIdx.y * 1024 + i], h[i]);
}

// per image: the three 256-entry look-up tables of the LUT ops (identity for the others) and the Contrast mean
__global__ __launch_bounds__(64) void aug_plan_kernel(const AugImage *__restrict__ imgs, const unsigned *__restrict__ hist,
                                                     uint8_t *__restrict__ lut, int *__restrict__ mean_out)
{
    const AugImage im = imgs[blockIdx.x];
    const unsigned *hb = hist + (size_t)blockIdx.x * 1024;
    const int c = threadIdx.x;
    if (c == 3) {                                                   // ImageEnhance.Contrast: int(mean(L) + 0.5), mean in double
        long long s = 0, n = 0;
        for (int i = 0; i < 256; i++) { s += (long long)i * hb[768 + i]; n += hb[768 + i]; }
        mean_out[blockIdx.x] = (int)((double)s / (double)n + 0.5);
        return;
    }
    if (c > 3) return;
    const unsigned *h = hb + c * 256;
    uint8_t *l = lut + ((size_t)blockIdx.x * 3 + c) * 256;
    for (int i = 0; i < 256; i++) l[i] = (uint8_t)i;
    if (im.op == 1) {                                               // ImageOps.autocontrast(cutoff=0)
        int lo = 0, hi = 255;
        while (lo < 256 && !h[lo]) lo++;
        while (hi >= 0 && !h[hi]) hi--;
        if (hi > lo) {
            const double scale = 255.0 / (double)(hi - lo);
            const double offset = -(double)lo * scale;
            for (int i = 0; i < 256; i++) {
                int v = (int)((double)i * scale + offset);
                l[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    } else if (im.op == 2) {                                        // ImageOps.equalize
        long long total = 0;
        int nonzero = 0, last = 0;
        for (int i = 0; i < 256; i++)
            if (h[i]) { total += h[i]; nonzero++; last = i; }
        if (nonzero > 1) {
            const long long step = (total - h[last]) / 255;
            if (step) {
                long long n = step / 2;
                for (int i = 0; i < 256; i++) {
                    const long long v = n / step;
                    l[i] = (uint8_t)(v > 255 ? 255 : v);
                    n += h[i];
                }
            }
        }
    } else if (im.op == 3) {                                        // mmcv.solarize(img, min(int(m * 256 / 10), 255))
        int thr = (im.magnitude * 256) / 10;
        thr = thr > 255 ? 255 : thr;
        for (int i = 0; i < 256; i++) l[i] = (uint8_t)(i < thr ? i : 255 - i);
    } else if (im.op == 8) {                                        // ImageOps.posterize(img, 4 - int(m * 4 / 10))
        const int bits = 4 - (im.magnitude * 4) / 10;
        const int mask = ~((1 << (8 - bits)) - 1) & 0xff;
        for (int i = 0; i < 256; i++) l[i] = (uint8_t)(i & mask);
    }
}

__device__ __forceinline__ uint8_t blend8(int d, int x, float alpha)
{
    const float prod = alpha * (float)(x - d);
    const float t = (float)d + prod;
    if (alpha >= 0.0f && alpha <= 1.0f) return (uint8_t)(int)t;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (uint8_t)(int)t);
}

__device__ __forceinline__ float norm1(int v, float mean, float sd)
{
    const float q = (float)v / 255.0f;
    const float r = q - mean;
    return r / sd;
}

// strong op + ToTensor/Normalize of both views:  weak u8 [B,S,S,3] -> wimg, simg float32 [B,3,S,S] (+ strong u8, optional)
__global__ __launch_bounds__(256) void aug_apply_kernel(const AugImage *__restrict__ imgs, const uint8_t *__restrict__ weak,
                                                       const uint8_t *__restrict__ lut, const int *__restrict__ mean_l,
                                                       float *__restrict__ wimg, float *__restrict__ simg,
                                                       uint8_t *__restrict__ strong_u8, int S)
{
    const AugImage im = imgs[blockIdx.z];
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
    if (X >= S) return;
    const uint8_t *img = weak + (size_t)blockIdx.z * S * S * 3;
    const uint8_t *p = img + ((size_t)Y * S + X) * 3;
    const int v[3] = {p[0], p[1], p[2]};
    int s[3] = {v[0], v[1], v[2]};
    const int op = im.op;
    if (op == 1 || op == 2 || op == 3 || op == 8) {
        const uint8_t *l = lut + (size_t)blockIdx.z * 768;
        s[0] = l[v[0]]; s[1] = l[256 + v[1]]; s[2] = l[512 + v[2]];
    } else if (op == 4) {                                           // Color: blend(luma, image)
        const int d = luma601(v[0], v[1], v[2]);
        for (int c = 0; c < 3; c++) s[c] = blend8(d, v[c], im.alpha);
    } else if (op == 5) {                                           // Contrast: blend(mean luma, image)
        const int d = mean_l[blockIdx.z];
        for (int c = 0; c < 3; c++) s[c] = blend8(d, v[c], im.alpha);
    } else if (op == 6) {                                           // Brightness: blend(black, image)
        for (int c = 0; c < 3; c++) s[c] = blend8(0, v[c], im.alpha);
    } else if (op == 7) {                                           // Sharpness: blend(SMOOTH(image), image); border pixels of SMOOTH are copies
        int d[3] = {v[0], v[1], v[2]};
        if (S >= 3 && X > 0 && X < S - 1 && Y > 0 && Y < S - 1) {
            const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
            for (int c = 0; c < 3; c++) {
                float acc = 0.5f;
                for (int dy = 1; dy >= -1; dy--) {                  // Pillow's order: row y+1, y, y-1
                    const uint8_t *r = img + ((size_t)(Y + dy) * S + X) * 3 + c;
                    float t = (float)r[-3] * k1;
                    t = t + (float)r[0] * (dy == 0 ? k5 : k1);
                    t = t + (float)r[3] * k1;
                    acc = acc + t;
                }
                d[c] = acc <= 0.0f ? 0 : (acc >= 255.0f ? 255 : (int)acc);
            }
        }
        for (int c = 0; c < 3; c++) s[c] = blend8(d[c], v[c], im.alpha);
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
    const size_t plane = (size_t)S * S, o = (size_t)blockIdx.z * 3 * plane + (size_t)Y * S + X;
    for (int c = 0; c < 3; c++) {
        wimg[o + c * plane] = norm1(v[c], mean[c], sd[c]);
        simg[o + c * plane] = norm1(s[c], mean[c], sd[c]);
    }
    if (strong_u8) {
        uint8_t *q = strong_u8 + ((size_t)blockIdx.z * plane + (size_t)Y * S + X) * 3;
        q[0] = (uint8_t)s[0]; q[1] = (uint8_t)s[1]; q[2] = (uint8_t)s[2];
    }
}

}  // namespace
}  // namespace cosa

using namespace cosa;

extern "C" int cosa_augment_record_bytes(void) { return (int)sizeof(AugImage); }

// workspace: T [B][max_rows][S][3] | crop [B][S][S][3] | tmp (same) | weak (same, when the caller does not ask for it) | hist | lut | mean
extern "C" size_t cosa_augment_workspace_bytes(int B, int S, int max_rows)
{
    const size_t img = align_up((size_t)B * S * S * 3, 256);
    return align_up((size_t)B * max_rows * S * 3, 256) + 3 * img + align_up((size_t)B * 1024 * sizeof(unsigned), 256) +
           align_up((size_t)B * 768, 256) + align_up((size_t)B * sizeof(int), 256);
}

extern "C" int cosa_augment_batch(const uint8_t *raw, const void *records, int B, int S, int max_rows, float *wimg, float *simg,
                                  uint8_t *crop_u8, uint8_t *weak_u8, uint8_t *strong_u8, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    COSA_REQUIRE(raw && records && wimg && simg && workspace, "cosa_augment_batch: null pointer");
    COSA_REQUIRE(B > 0 && B <= 65535 && S > 0 && S <= 65535 && max_rows > 0 && max_rows <= 65535, "cosa_augment_batch: bad shape");
    if (workspace_bytes < cosa_augment_workspace_bytes(B, S, max_rows)) {
        set_error("cosa_augment_batch: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    const AugImage *imgs = static_cast<const AugImage *>(records);
    Carver cv(workspace);
    uint8_t *T = cv.take<uint8_t>((size_t)B * max_rows * S * 3);
    uint8_t *crop = cv.take<uint8_t>((size_t)B * S * S * 3);
    uint8_t *tmp = cv.take<uint8_t>((size_t)B * S * S * 3);
    uint8_t *weak = cv.take<uint8_t>((size_t)B * S * S * 3);
    unsigned *hist = cv.take<unsigned>((size_t)B * 1024);
    uint8_t *lut = cv.take<uint8_t>((size_t)B * 768);
    int *mean_l = cv.take<int>((size_t)B);
    if (crop_u8) crop = crop_u8;
    if (weak_u8) weak = weak_u8;
    const dim3 blk(256), gx((S + 255) / 256, S, B);
    hipLaunchKernelGGL(aug_hpass_kernel, dim3((S + 255) / 256, max_rows, B), blk, 0, st, raw, imgs, T, S, max_rows);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_vpass_kernel, gx, blk, 0, st, imgs, T, crop, S, max_rows);
    COSA_LAUNCH_CHECK();
    // GaussianBlur: three passes along x, then three along y (images that drew no blur are copied through)
    const uint8_t *src = crop;
    uint8_t *ping[2] = {tmp, weak};
    for (int pass = 0; pass < 6; pass++) {
        uint8_t *dst = ping[pass & 1];                               // ends in `weak` (pass 5 -> ping[1])
        hipLaunchKernelGGL(aug_box_pass_kernel, gx, blk, 0, st, imgs, src, dst, S, pass < 3 ? 1 : 0);
        COSA_LAUNCH_CHECK();
        src = dst;
    }
    COSA_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)B * 1024 * sizeof(unsigned), st));
    hipLaunchKernelGGL(aug_hist_kernel, dim3(32, B), blk, 0, st, weak, hist, S);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_plan_kernel, dim3(B), dim3(64), 0, st, imgs, hist, lut, mean_l);
    COSA_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_apply_kernel, gx, blk, 0, st, imgs, weak, lut, mean_l, wimg, simg, strong_u8, S);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
