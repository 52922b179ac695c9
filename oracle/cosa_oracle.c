/*
 * cosa_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the integer/index-producing and HBM-bound stages of
 * CoSA's per-iteration hot path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product path
 * (cosa_amd/) never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to the reference repo
 * root) whose behaviour it restates.  The arithmetic is written as an explicit
 * operation-order SPEC (see DESIGN.md "Arithmetic spec"): every float op is a
 * single IEEE-754 binary32 operation in the order written, fused multiply-adds
 * appear only where fmaf() is spelled out, and the file is compiled with
 * -ffp-contract=off.  The HIP kernels follow the same spec independently, so
 * label maps are bit-identical between this oracle and the GPU.
 *
 * Parity pinning: the reference ships no tests.  This oracle is
 * pinned against (1) the reference's own Python modules imported in the
 * authoring container and (2) the reference's C++ bilateral filter compiled
 * from its sources into oracle/_ref/ -- see oracle/gen_golden.py and
 * tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Deterministic expf (spec E).  Cody-Waite reduction + degree-5 minimax       */
/* (Cephes coefficients), all steps spelled as fmaf/mul/add.  |err| < 1 ulp.   */
/* Inputs below -87 return exactly 0 (no subnormal results on either side).    */
/* ------------------------------------------------------------------------- */
static inline float orc_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float k = rintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float r2 = r * r;
    float e = fmaf(p, r2, r);
    e = e + 1.0f;
    return ldexpf(e, (int)k);
}

float orc_expf_export(float x) { return orc_expf(x); }

/* ------------------------------------------------------------------------- */
/* denormalize_img: utils/torch_helper.py:354-367                             */
/* x*std+mean -> uint8 (C truncation toward zero, wraps mod 256 like torch's  */
/* float->uint8 cast on x86 for in-range values) -> /255                      */
/* ------------------------------------------------------------------------- */
void orc_denormalize_img(const float *img, float *out, int B, int H, int W)
{
    static const float mean[3] = {123.675f, 116.28f, 103.53f};
    static const float std_[3] = {58.395f, 57.12f, 57.375f};
    size_t hw = (size_t)H * W;
    for (int b = 0; b < B; b++)
        for (int c = 0; c < 3; c++) {
            const float *s = img + ((size_t)b * 3 + c) * hw;
            float *d = out + ((size_t)b * 3 + c) * hw;
            for (size_t i = 0; i < hw; i++) {
                float v = s[i] * std_[c];
                v = v + mean[c];
                int iv = (int)v;               /* truncation */
                uint8_t u = (uint8_t)iv;       /* wrap */
                d[i] = (float)u / 255.0f;
            }
        }
}

/* ------------------------------------------------------------------------- */
/* CAM min-max normalisation tail of multi_scale_camseg:                      */
/* utils/seg_helper.py:264-270                                                */
/*   cam = cam + maxpool(-cam)  ==  cam - min ;  cam /= max(cam) + 1e-5       */
/* in place on [BC, HW] planes.                                               */
/* ------------------------------------------------------------------------- */
void orc_cam_minmax_norm(float *cam, int BC, int HW)
{
    for (int p = 0; p < BC; p++) {
        float *x = cam + (size_t)p * HW;
        float mneg = -x[0];
        for (int i = 1; i < HW; i++) { float v = -x[i]; if (v > mneg) mneg = v; }
        float mx = x[0] + mneg;
        for (int i = 0; i < HW; i++) { float v = x[i] + mneg; x[i] = v; if (v > mx) mx = v; }
        float den = mx + 1e-5f;
        for (int i = 0; i < HW; i++) x[i] = x[i] / den;
    }
}

/* ------------------------------------------------------------------------- */
/* Active-channel list: cls_labels_with_bkg nonzero -> current_labels         */
/* utils/seg_helper.py:755-765                                                */
/* act[0]=0 (background), then c+1 for every c with label[c] != 0.            */
/* ------------------------------------------------------------------------- */
static int orc_active(const float *label, int C, int *act)
{
    int K = 0;
    act[K++] = 0;
    for (int c = 0; c < C; c++) if (label[c] != 0.0f) act[K++] = c + 1;
    return K;
}

/* ------------------------------------------------------------------------- */
/* Spec L: threshold-plane cat + bilinear /2 + active-channel softmax.        */
/* utils/seg_helper.py:739-753 (cat + interpolate), :766-767 (softmax),       */
/* with cam_validation (:547-551) folded in (cam * label).                    */
/* cam: [C,S,S] of ONE image, label: [C].  out: [K, s, s] (s = S/ds or S).    */
/* ------------------------------------------------------------------------- */
static void orc_lowres_softmax(const float *cam, const float *label, int C, int S,
                               int downscale, float thr, const int *act, int K, float *out)
{
    int s = downscale ? S / downscale : S;
    size_t SS = (size_t)S * S, ss = (size_t)s * s;
    float *v = (float *)malloc(sizeof(float) * K);
    for (int y = 0; y < s; y++)
        for (int x = 0; x < s; x++) {
            for (int k = 0; k < K; k++) {
                if (k == 0) { v[k] = thr; continue; }
                int c = act[k] - 1;
                const float *pl = cam + (size_t)c * SS;
                float lab = label[c];
                if (downscale) {
                    /* exact x2 case of ATen bilinear (align_corners=False): 0.5/0.5 taps */
                    float a00 = pl[(size_t)(2 * y) * S + 2 * x] * lab;
                    float a01 = pl[(size_t)(2 * y) * S + 2 * x + 1] * lab;
                    float a10 = pl[(size_t)(2 * y + 1) * S + 2 * x] * lab;
                    float a11 = pl[(size_t)(2 * y + 1) * S + 2 * x + 1] * lab;
                    float r0 = a00 * 0.5f + a01 * 0.5f;
                    float r1 = a10 * 0.5f + a11 * 0.5f;
                    v[k] = r0 * 0.5f + r1 * 0.5f;
                } else {
                    v[k] = pl[(size_t)y * S + x] * lab;
                }
            }
            float m = v[0];
            for (int k = 1; k < K; k++) if (v[k] > m) m = v[k];
            float sum = 0.0f;
            for (int k = 0; k < K; k++) { v[k] = orc_expf(v[k] - m); sum = sum + v[k]; }
            for (int k = 0; k < K; k++) out[(size_t)k * ss + (size_t)y * s + x] = v[k] / sum;
        }
    free(v);
}

/* ------------------------------------------------------------------------- */
/* PAR: models/PAR.py:26-91.  img [3,s,s] in [0,1], masks [K,s,s] (same size, */
/* so the align_corners=True resize at :66 is the identity).                  */
/* Neighbour order n = di*8 + t, t -> (dy,dx)*d from get_kernel (:10-24).     */
/* ------------------------------------------------------------------------- */
static const int PAR_DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
static const int PAR_DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* position prior: softmax_n( -(pos_n/(std(pos)+1e-8)/w1)^2 ), models/PAR.py:51-62,77,82-85 */
void orc_par_pos_weights(const int *dil, int ND, float *posw)
{
    int NN = ND * 8;
    float *pos = (float *)malloc(sizeof(float) * NN);
    const float sq2 = (float)sqrt(2.0);
    for (int di = 0; di < ND; di++)
        for (int t = 0; t < 8; t++) {
            float kk = (t == 0 || t == 2 || t == 5 || t == 7) ? sq2 : 1.0f;
            pos[di * 8 + t] = kk * (float)dil[di];
        }
    float sum = 0.0f;
    for (int n = 0; n < NN; n++) sum = sum + pos[n];
    float mean = sum / (float)NN;
    float var = 0.0f;
    for (int n = 0; n < NN; n++) { float dlt = pos[n] - mean; var = var + dlt * dlt; }
    var = var / (float)(NN - 1);
    float sd = sqrtf(var);
    float mx = -INFINITY;
    for (int n = 0; n < NN; n++) {
        float q = pos[n] / (sd + 1e-8f);
        q = q / 0.3f;
        pos[n] = -(q * q);
        if (pos[n] > mx) mx = pos[n];
    }
    float es = 0.0f;
    for (int n = 0; n < NN; n++) { posw[n] = orc_expf(pos[n] - mx); es = es + posw[n]; }
    for (int n = 0; n < NN; n++) posw[n] = posw[n] / es;
    free(pos);
}

/* affinity for one image: aff [NN, s, s]   (models/PAR.py:69-85) */
void orc_par_affinity(const float *img, int h, int w, const int *dil, int ND, float *aff)
{
    int NN = ND * 8;
    size_t hw = (size_t)h * w;
    float *posw = (float *)malloc(sizeof(float) * NN);
    float *lg = (float *)malloc(sizeof(float) * NN);
    orc_par_pos_weights(dil, ND, posw);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sd[3];
            for (int c = 0; c < 3; c++) {
                const float *pl = img + (size_t)c * hw;
                float sum = 0.0f;
                for (int n = 0; n < NN; n++) {
                    int d = dil[n >> 3], t = n & 7;
                    int yy = clampi(y + PAR_DY[t] * d, 0, h - 1), xx = clampi(x + PAR_DX[t] * d, 0, w - 1);
                    sum = sum + pl[(size_t)yy * w + xx];
                }
                float mean = sum / (float)NN;
                float var = 0.0f;
                for (int n = 0; n < NN; n++) {
                    int d = dil[n >> 3], t = n & 7;
                    int yy = clampi(y + PAR_DY[t] * d, 0, h - 1), xx = clampi(x + PAR_DX[t] * d, 0, w - 1);
                    float dlt = pl[(size_t)yy * w + xx] - mean;
                    var = var + dlt * dlt;
                }
                var = var / (float)(NN - 1);
                sd[c] = sqrtf(var) + 1e-8f;
            }
            float mx = -INFINITY;
            for (int n = 0; n < NN; n++) {
                int d = dil[n >> 3], t = n & 7;
                int yy = clampi(y + PAR_DY[t] * d, 0, h - 1), xx = clampi(x + PAR_DX[t] * d, 0, w - 1);
                float acc = 0.0f;
                for (int c = 0; c < 3; c++) {
                    const float *pl = img + (size_t)c * hw;
                    float a = fabsf(pl[(size_t)yy * w + xx] - pl[(size_t)y * w + x]);
                    float q = a / sd[c];
                    q = q / 0.3f;
                    acc = acc + (-(q * q));
                }
                lg[n] = acc / 3.0f;
                if (lg[n] > mx) mx = lg[n];
            }
            float es = 0.0f;
            for (int n = 0; n < NN; n++) { lg[n] = orc_expf(lg[n] - mx); es = es + lg[n]; }
            for (int n = 0; n < NN; n++) {
                float a = lg[n] / es;
                aff[(size_t)n * hw + (size_t)y * w + x] = a + 0.01f * posw[n];
            }
        }
    free(posw);
    free(lg);
}

/* T propagation steps on K planes (models/PAR.py:87-89); result in `masks`. */
void orc_par_propagate(const float *aff, float *masks, int K, int h, int w, const int *dil, int ND, int T)
{
    int NN = ND * 8;
    size_t hw = (size_t)h * w;
    float *tmp = (float *)malloc(sizeof(float) * K * hw);
    float *src = masks, *dst = tmp;
    for (int it = 0; it < T; it++) {
        for (int k = 0; k < K; k++) {
            const float *pl = src + (size_t)k * hw;
            float *o = dst + (size_t)k * hw;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    float acc = 0.0f;
                    for (int n = 0; n < NN; n++) {
                        int d = dil[n >> 3], t = n & 7;
                        int yy = clampi(y + PAR_DY[t] * d, 0, h - 1), xx = clampi(x + PAR_DX[t] * d, 0, w - 1);
                        acc = acc + pl[(size_t)yy * w + xx] * aff[(size_t)n * hw + (size_t)y * w + x];
                    }
                    o[(size_t)y * w + x] = acc;
                }
        }
        float *sw = src; src = dst; dst = sw;
    }
    if (src != masks) memcpy(masks, src, sizeof(float) * K * hw);
    free(tmp);
}

/* PAR.forward for one image (same-size masks). */
void orc_par_forward(const float *img, float *masks, int K, int h, int w, const int *dil, int ND, int T)
{
    float *aff = (float *)malloc(sizeof(float) * ND * 8 * (size_t)h * w);
    orc_par_affinity(img, h, w, dil, ND, aff);
    orc_par_propagate(aff, masks, K, h, w, dil, ND, T);
    free(aff);
}

/* ------------------------------------------------------------------------- */
/* Spec U: bilinear up to (S,S) (align_corners=False) + first-max argmax +    */
/* valid_key gather.  utils/seg_helper.py:787-797.                            */
/* p: [K, s, s] -> lab [S*S] int32                                            */
/* Interpolation form (matches ATen CPU's contracted evaluation):             */
/*   r0 = fma(p00, lx0, p01*lx1); r1 = fma(p10, lx0, p11*lx1);                */
/*   v  = fma(r0, ly0, r1*ly1)                                                */
/* ------------------------------------------------------------------------- */
static inline void orc_src_index(int dst, int in, int out, int *i0, int *i1, float *l0, float *l1)
{
    if (in == out) { *i0 = dst; *i1 = dst < in - 1 ? dst + 1 : dst; *l0 = 1.0f; *l1 = 0.0f; return; }
    float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int i = (int)src;
    if (i > in - 1) i = in - 1;
    *i0 = i;
    *i1 = i < in - 1 ? i + 1 : i;
    *l1 = src - (float)i;
    *l0 = 1.0f - *l1;
}

static void orc_upsample_argmax(const float *p, int K, int s, int S, const int *act, int *lab)
{
    size_t ss = (size_t)s * s;
    for (int Y = 0; Y < S; Y++) {
        int y0, y1; float ly0, ly1;
        orc_src_index(Y, s, S, &y0, &y1, &ly0, &ly1);
        for (int X = 0; X < S; X++) {
            int x0, x1; float lx0, lx1;
            orc_src_index(X, s, S, &x0, &x1, &lx0, &lx1);
            float best = 0.0f; int bi = 0;
            for (int k = 0; k < K; k++) {
                const float *pl = p + (size_t)k * ss;
                float r0 = fmaf(pl[(size_t)y0 * s + x0], lx0, pl[(size_t)y0 * s + x1] * lx1);
                float r1 = fmaf(pl[(size_t)y1 * s + x0], lx0, pl[(size_t)y1 * s + x1] * lx1);
                float v = fmaf(r0, ly0, r1 * ly1);
                if (k == 0 || v > best) { best = v; bi = k; }
            }
            lab[(size_t)Y * S + X] = act[bi];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* cam2mask: utils/seg_helper.py:721-785 (+ _refine_cams :787-797).           */
/* cams [B,C,S,S] (raw, un-validated: cam_validation is folded in),           */
/* labels [B,C], boxes [B,4] int32 (h0,h1,w0,w1), images [B,3,S,S] in [0,1]   */
/* (only read when use_par), mask out [B,S,S] float {0..C,255}.               */
/* ------------------------------------------------------------------------- */
static void orc_resize_half_image(const float *img, int S, float *out)
{
    int s = S / 2;
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < s; y++)
            for (int x = 0; x < s; x++) {
                const float *pl = img + (size_t)c * S * S;
                float r0 = pl[(size_t)(2 * y) * S + 2 * x] * 0.5f + pl[(size_t)(2 * y) * S + 2 * x + 1] * 0.5f;
                float r1 = pl[(size_t)(2 * y + 1) * S + 2 * x] * 0.5f + pl[(size_t)(2 * y + 1) * S + 2 * x + 1] * 0.5f;
                out[(size_t)c * s * s + (size_t)y * s + x] = r0 * 0.5f + r1 * 0.5f;
            }
}

void orc_cam2mask(const float *images, const int *boxes, const float *cams, const float *labels,
                  int B, int C, int S, float thr_hi, float thr_lo, int downscale,
                  int use_par, const int *dil, int ND, int T, float ignore_index, float *mask)
{
    int s = downscale ? S / downscale : S;
    size_t SS = (size_t)S * S, ss = (size_t)s * s;
    int *act = (int *)malloc(sizeof(int) * (C + 1));
    float *phi = (float *)malloc(sizeof(float) * (C + 1) * ss);
    float *plo = (float *)malloc(sizeof(float) * (C + 1) * ss);
    int *lhi = (int *)malloc(sizeof(int) * SS);
    int *llo = (int *)malloc(sizeof(int) * SS);
    float *simg = (float *)malloc(sizeof(float) * 3 * ss);
    for (int b = 0; b < B; b++) {
        const float *cam = cams + (size_t)b * C * SS;
        const float *lab = labels + (size_t)b * C;
        int K = orc_active(lab, C, act);
        orc_lowres_softmax(cam, lab, C, S, downscale, thr_hi, act, K, phi);
        orc_lowres_softmax(cam, lab, C, S, downscale, thr_lo, act, K, plo);
        if (use_par) {
            const float *im = images + (size_t)b * 3 * SS;
            if (downscale) { orc_resize_half_image(im, S, simg); im = simg; }
            float *aff = (float *)malloc(sizeof(float) * ND * 8 * ss);
            orc_par_affinity(im, s, s, dil, ND, aff);
            orc_par_propagate(aff, phi, K, s, s, dil, ND, T);
            orc_par_propagate(aff, plo, K, s, s, dil, ND, T);
            free(aff);
        }
        orc_upsample_argmax(phi, K, s, S, act, lhi);
        orc_upsample_argmax(plo, K, s, S, act, llo);
        int h0 = boxes[b * 4 + 0], h1 = boxes[b * 4 + 1], w0 = boxes[b * 4 + 2], w1 = boxes[b * 4 + 3];
        float *m = mask + (size_t)b * SS;
        for (int Y = 0; Y < S; Y++)
            for (int X = 0; X < S; X++) {
                int in = (Y >= h0 && Y < h1 && X >= w0 && X < w1);
                float hi = in ? (float)lhi[(size_t)Y * S + X] : ignore_index;
                float lo = in ? (float)llo[(size_t)Y * S + X] : ignore_index;
                float r = hi;
                if (hi == 0.0f) r = ignore_index;
                if (hi + lo == 0.0f) r = 0.0f;
                m[(size_t)Y * S + X] = r;
            }
    }
    free(act); free(phi); free(plo); free(lhi); free(llo); free(simg);
}

/* expose the low-res softmax for stage-wise tests */
int orc_lowres_softmax_image(const float *cam, const float *label, int C, int S, int downscale,
                             float thr, float *out /* [(C+1), s, s] */, int *act_out)
{
    int K = orc_active(label, C, act_out);
    orc_lowres_softmax(cam, label, C, S, downscale, thr, act_out, K, out);
    return K;
}

/* ------------------------------------------------------------------------- */
/* Permutohedral lattice bilateral filter.                                    */
/* utils/bilateralfilter/bilateralfilter.cpp:4-55 (features, batch driver),   */
/* utils/bilateralfilter/permutohedral.cpp:115-297 (init, SSE path:           */
/* round-to-nearest-even, blocks of 4 zero-padded) and :507-571 (compute).    */
/* Own data structures: 64-bit packed keys in an open-addressing table;       */
/* lattice ids are assigned in first-touch order like the reference, so the   */
/* splat summation order -- and therefore every output bit -- is the same.    */
/* ------------------------------------------------------------------------- */
/* The algorithm is dimension-generic.  ORC_PD = 5 (default) is the bilateral filter of the hot path: positions / sigma_xy and colours /
 * sigma_rgb, pinned bit for bit by the reference's own C++ (tests/golden/bilateral.npz, oracle/_ref).  The SAME code built with -DORC_PD=2
 * (liboracle_d2.so) is the position-only Gaussian kernel of the dense-CRF post-processing (utils/seg_helper.py:961-996 via pydensecrf's
 * addPairwiseGaussian): features x / sxy, y / sxy.                                                                                  */
#ifndef ORC_PD
#define ORC_PD 5
#endif
#define PD ORC_PD       /* feature dimension */
#define PD1 (ORC_PD + 1)

typedef struct {
    int N, M;
    int *offset;        /* [N][6] lattice id */
    float *bary;        /* [N][6] */
    int *nb;            /* [6][M][2] blur neighbours (-1 = none) */
} orc_lattice;

typedef struct { uint64_t *keys; int *ids; size_t cap; } orc_hash;

static inline uint64_t orc_pack(const short *k)
{
    uint64_t r = 0;
    for (int i = 0; i < PD; i++) r = (r << 12) | (uint64_t)((uint16_t)(k[i] + 2048) & 0xFFF);
    return r | (1ull << 63);       /* never zero */
}
static inline size_t orc_hmix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; return (size_t)k; }

static int orc_hfind(orc_hash *h, const short *key, int create, int *count, short *keystore)
{
    uint64_t pk = orc_pack(key);
    size_t i = orc_hmix(pk) & (h->cap - 1);
    for (;;) {
        if (h->keys[i] == 0) {
            if (!create) return -1;
            h->keys[i] = pk; h->ids[i] = *count;
            memcpy(keystore + (size_t)(*count) * PD, key, sizeof(short) * PD);
            return (*count)++;
        }
        if (h->keys[i] == pk) return h->ids[i];
        i = (i + 1) & (h->cap - 1);
    }
}

/* returns 0 on success, -1 when a key component leaves the packable range */
int orc_lattice_init(orc_lattice *L, const float *image /* CHW 0..255 */, int H, int W, float sigmargb, float sigmaxy)
{
    int N = H * W, Npad = (N + 3) & ~3;
    L->N = N;
    L->offset = (int *)malloc(sizeof(int) * (size_t)Npad * PD1);
    L->bary = (float *)malloc(sizeof(float) * (size_t)Npad * PD1);
    orc_hash ht; ht.cap = 1; while (ht.cap < (size_t)Npad * PD1 * 2) ht.cap <<= 1;
    ht.keys = (uint64_t *)calloc(ht.cap, sizeof(uint64_t));
    ht.ids = (int *)malloc(sizeof(int) * ht.cap);
    short *keystore = (short *)malloc(sizeof(short) * PD * (size_t)Npad * PD1);
    int count = 0, bad = 0;

    float scale[PD];
    double inv_std_dev = sqrt(2.0 / 3.0) * (PD + 1);
    for (int i = 0; i < PD; i++) scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)(float)inv_std_dev);
    const float inv6 = 1.0f / (float)PD1;
    short canonical[PD1][PD1];
    for (int i = 0; i <= PD; i++) {
        for (int j = 0; j <= PD - i; j++) canonical[i][j] = (short)i;
        for (int j = PD - i + 1; j <= PD; j++) canonical[i][j] = (short)(i - PD1);
    }
    size_t hw = (size_t)H * W;
    for (int p = 0; p < Npad; p++) {
        float f[PD];
        for (int i = 0; i < PD; i++) f[i] = 0.0f;
        if (p < N) {
            int xi = p % W, yj = p / W;
            f[0] = (float)xi / sigmaxy;
            f[1] = (float)yj / sigmaxy;
#if ORC_PD == 5
            f[2] = image[p] / sigmargb;
            f[3] = image[hw + p] / sigmargb;
            f[4] = image[2 * hw + p] / sigmargb;
#endif
        }
        float el[PD1], rem0[PD1], rank[PD1], bc[PD1 + 1];
        float sm = 0.0f;
        for (int j = PD; j > 0; j--) {
            float cf = f[j - 1] * scale[j - 1];
            el[j] = sm - (float)j * cf;
            sm = sm + cf;
        }
        el[0] = sm;
        float sum = 0.0f;
        for (int i = 0; i <= PD; i++) {
            float v = inv6 * el[i];
            v = rintf(v);                       /* nearest-even, as cvtps_epi32 */
            rem0[i] = v * (float)PD1;
            sum = sum + v;
        }
        for (int i = 0; i <= PD; i++) rank[i] = 0.0f;
        for (int i = 0; i < PD; i++) {
            float di = el[i] - rem0[i];
            for (int j = i + 1; j <= PD; j++) {
                float dj = el[j] - rem0[j];
                if (di < dj) rank[i] = rank[i] + 1.0f; else rank[j] = rank[j] + 1.0f;
            }
        }
        for (int i = 0; i <= PD; i++) {
            rank[i] = rank[i] + sum;
            if (rank[i] < 0.0f) { rank[i] = rank[i] + (float)PD1; rem0[i] = rem0[i] + (float)PD1; }
            else if (rank[i] >= (float)PD1) { rank[i] = rank[i] - (float)PD1; rem0[i] = rem0[i] - (float)PD1; }
        }
        for (int i = 0; i <= PD + 1; i++) bc[i] = 0.0f;
        for (int i = 0; i <= PD; i++) {
            float v = (el[i] - rem0[i]) * inv6;
            int q = PD - (int)rank[i];
            bc[q] = bc[q] + v;
            bc[q + 1] = bc[q + 1] - v;
        }
        bc[0] = bc[0] + (1.0f + bc[PD + 1]);
        for (int r = 0; r <= PD; r++) {
            short key[PD];
            for (int i = 0; i < PD; i++) {
                float kv = rem0[i] + (float)canonical[r][(int)rank[i]];
                if (kv < -2048.0f || kv > 2047.0f) bad = 1;
                key[i] = (short)kv;
            }
            L->offset[(size_t)p * PD1 + r] = orc_hfind(&ht, key, 1, &count, keystore);
            L->bary[(size_t)p * PD1 + r] = bc[r];
        }
    }
    int M = count;
    L->M = M;
    L->nb = (int *)malloc(sizeof(int) * 2 * (size_t)PD1 * (M > 0 ? M : 1));
    for (int j = 0; j <= PD; j++)
        for (int i = 0; i < M; i++) {
            const short *key = keystore + (size_t)i * PD;
            short n1[PD], n2[PD];
            int ok1 = 1, ok2 = 1;
            for (int k = 0; k < PD; k++) { n1[k] = (short)(key[k] - 1); n2[k] = (short)(key[k] + 1); }
            if (j < PD) { n1[j] = (short)(key[j] + PD); n2[j] = (short)(key[j] - PD); }
            for (int k = 0; k < PD; k++) { if (n1[k] < -2048 || n1[k] > 2047) ok1 = 0; if (n2[k] < -2048 || n2[k] > 2047) ok2 = 0; }
            L->nb[((size_t)j * M + i) * 2 + 0] = ok1 ? orc_hfind(&ht, n1, 0, &count, keystore) : -1;
            L->nb[((size_t)j * M + i) * 2 + 1] = ok2 ? orc_hfind(&ht, n2, 0, &count, keystore) : -1;
        }
    free(ht.keys); free(ht.ids); free(keystore);
    return bad ? -1 : 0;
}

void orc_lattice_free(orc_lattice *L) { free(L->offset); free(L->bary); free(L->nb); }

/* one scalar channel: splat, 6 blur passes, slice.  permutohedral.cpp:507-571 (value_size=1) */
void orc_lattice_compute(const orc_lattice *L, const float *in, float *out)
{
    int N = L->N, M = L->M;
    float *val = (float *)calloc((size_t)M + 2, sizeof(float));
    float *nval = (float *)calloc((size_t)M + 2, sizeof(float));
    for (int i = 0; i < N; i++)
        for (int j = 0; j <= PD; j++) {
            int o = L->offset[(size_t)i * PD1 + j] + 1;
            float w = L->bary[(size_t)i * PD1 + j];
            val[o] = val[o] + w * in[i];
        }
    for (int j = 0; j <= PD; j++) {
        for (int i = 0; i < M; i++) {
            int n1 = L->nb[((size_t)j * M + i) * 2 + 0] + 1;
            int n2 = L->nb[((size_t)j * M + i) * 2 + 1] + 1;
            nval[i + 1] = val[i + 1] + 0.5f * (val[n1] + val[n2]);
        }
        float *t = val; val = nval; nval = t;
    }
    float alpha = 1.0f / (1.0f + powf(2.0f, -(float)PD));
    for (int i = 0; i < N; i++) {
        float acc = 0.0f;
        for (int j = 0; j <= PD; j++) {
            int o = L->offset[(size_t)i * PD1 + j] + 1;
            float w = L->bary[(size_t)i * PD1 + j] * alpha;
            acc = acc + w * val[o];
        }
        out[i] = acc;
    }
    free(val); free(nval);
}

/* bilateralfilter_batch: utils/bilateralfilter/bilateralfilter.cpp:42-55 (serial over n here) */
int orc_bilateralfilter_batch(const float *images, const float *ins, float *outs,
                              int N, int K, int H, int W, float sigmargb, float sigmaxy, int *M_out)
{
    int rc = 0;
    size_t hw = (size_t)H * W;
    for (int n = 0; n < N; n++) {
        orc_lattice L;
        if (orc_lattice_init(&L, images + (size_t)n * 3 * hw, H, W, sigmargb, sigmaxy)) rc = -1;
        if (M_out) M_out[n] = L.M;
        for (int k = 0; k < K; k++)
            orc_lattice_compute(&L, ins + ((size_t)n * K + k) * hw, outs + ((size_t)n * K + k) * hw);
        orc_lattice_free(&L);
    }
    return rc;
}

/* ------------------------------------------------------------------------- */
/* DenseEnergyLossFunction forward core: utils/seg_helper.py:867-896.         */
/* images [N,3,H,W] 0..255, seg [N,K,H,W] probabilities, roi [N,H,W],         */
/* unlabel [N,H,W] (uint8 0/1).  Writes AS (gated) [N,K,H,W]; returns loss.   */
/* ------------------------------------------------------------------------- */
float orc_dense_energy_forward(const float *images, const float *seg, const float *roi, const uint8_t *unlabel,
                               int N, int K, int H, int W, float sigmargb, float sigmaxy, float *AS)
{
    size_t hw = (size_t)H * W;
    float *segm = (float *)malloc(sizeof(float) * (size_t)N * K * hw);
    float *gate = (float *)malloc(sizeof(float) * (size_t)N * hw);
    for (int n = 0; n < N; n++)
        for (size_t p = 0; p < hw; p++) {
            float mx = seg[((size_t)n * K) * hw + p];
            for (int k = 1; k < K; k++) { float v = seg[((size_t)n * K + k) * hw + p]; if (v > mx) mx = v; }
            float g = roi[n * hw + p] - mx;
            if (unlabel[n * hw + p]) g = 1.0f;
            if (g < 0.0f) g = 0.0f;
            gate[n * hw + p] = g;
            for (int k = 0; k < K; k++) segm[((size_t)n * K + k) * hw + p] = seg[((size_t)n * K + k) * hw + p] * roi[n * hw + p];
        }
    orc_bilateralfilter_batch(images, segm, AS, N, K, H, W, sigmargb, sigmaxy, NULL);
    double acc = 0.0;
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++)
            for (size_t p = 0; p < hw; p++) {
                size_t i = ((size_t)n * K + k) * hw + p;
                AS[i] = AS[i] * gate[n * hw + p];
                acc += (double)segm[i] * (double)AS[i];
            }
    free(segm); free(gate);
    return (float)(-acc / (double)N);
}

/* ========================================================================= */
/* Evaluation path (SURVEY f-1): evaluation_engine.py:74-126,198-207,         */
/* utils/seg_helper.py:515-546 (cam_to_label), :581-591 (seg_validation),     */
/* utils/evaluation.py:10-70 (_fast_hist, scores, pseudo_scores).             */
/* ========================================================================= */

/* Spec R: F.interpolate(mode='bilinear', align_corners=False) to an arbitrary (H,W), as ATen's CPU kernel evaluates it for
 * outputs of image size (pinned by tests/golden/eval.npz at 333x500, 375x500, 500x281):
 *   scale = (float)in / (float)out;  src = max(fmaf(scale, dst + 0.5f, -0.5f), 0);  i0 = min((int)src, in-1);  l1 = src - i0
 *   r0 = fma(p00, lx0, p01*lx1); r1 = fma(p10, lx0, p11*lx1); v = fma(r0, ly0, r1*ly1)
 * (for the exact x2 / x0.5 / x16 ratios of the training path the fused source index equals orc_src_index's).           */
static inline void orc_src_index_r(int dst, int in, int out, int *i0, int *i1, float *l0, float *l1)
{
    float scale = (float)in / (float)out;
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (src < 0.0f) src = 0.0f;
    int i = (int)src;
    if (i > in - 1) i = in - 1;
    float lam = src - (float)i;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    *i0 = i;
    *i1 = i < in - 1 ? i + 1 : i;
    *l1 = lam;
    *l0 = 1.0f - lam;
}

static inline float orc_bilerp(const float *pl, int w, int y0, int y1, int x0, int x1, float ly0, float ly1, float lx0, float lx1)
{
    float r0 = fmaf(pl[(size_t)y0 * w + x0], lx0, pl[(size_t)y0 * w + x1] * lx1);
    float r1 = fmaf(pl[(size_t)y1 * w + x0], lx0, pl[(size_t)y1 * w + x1] * lx1);
    return fmaf(r0, ly0, r1 * ly1);
}

void orc_resize_bilinear(const float *p, int C, int h, int w, int H, int W, float *out)
{
    for (int c = 0; c < C; c++)
        for (int Y = 0; Y < H; Y++) {
            int y0, y1; float ly0, ly1;
            orc_src_index_r(Y, h, H, &y0, &y1, &ly0, &ly1);
            for (int X = 0; X < W; X++) {
                int x0, x1; float lx0, lx1;
                orc_src_index_r(X, w, W, &x0, &x1, &lx0, &lx1);
                out[((size_t)c * H + Y) * W + X] = orc_bilerp(p + (size_t)c * h * w, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
            }
        }
}

/* cam_to_label (seg_helper.py:515-546) on an already resized CAM [B,C,H,W].  cls_label may be NULL.  boxes (B x 4:
 * h0,h1,w0,w1; NULL = the `img_box is None` return) ; valid_cam (optional) receives cls_label * cam.                */
void orc_cam_to_label(const float *cam, const float *cls_label, int B, int C, int H, int W, float bkg_thre,
                      const int *boxes, int ignore_mid, float high_thre, float low_thre, int ignore_index,
                      int64_t *label, float *valid_cam)
{
    size_t hw = (size_t)H * W;
    for (int b = 0; b < B; b++)
        for (size_t p = 0; p < hw; p++) {
            float best = 0.0f; int bi = 0;
            for (int c = 0; c < C; c++) {
                float v = cam[((size_t)b * C + c) * hw + p];
                if (cls_label) v = cls_label[b * C + c] * v;
                if (valid_cam) valid_cam[((size_t)b * C + c) * hw + p] = v;
                if (c == 0 || v > best) { best = v; bi = c; }
            }
            int64_t l = bi + 1;
            if (best <= bkg_thre) l = 0;
            if (boxes) {
                if (ignore_mid) {
                    if (best <= high_thre) l = ignore_index;
                    if (best <= low_thre) l = 0;
                }
                int y = (int)(p / W), x = (int)(p % W);
                const int *bx = boxes + 4 * b;
                if (!(y >= bx[0] && y < bx[1] && x >= bx[2] && x < bx[3])) l = ignore_index;
            }
            label[(size_t)b * hw + p] = l;
        }
}

/* one evaluation image: resize the (S,S) CAM and segmentation logits to the ground truth's (H,W), then
 * cam_to_label(bkg_thre) and argmax of the raw / class-validated logits (evaluation_engine.py:96-126,198-200).
 * cam [C,S,S], seg [C+1,S,S], cls_label [C] */
void orc_eval_labels(const float *cam, const float *seg, const float *cls_label, int C, int S, int H, int W, float bkg_thre,
                     uint8_t *lab_cam, uint8_t *lab_ps, uint8_t *lab_vd)
{
    size_t ss = (size_t)S * S;
    for (int Y = 0; Y < H; Y++) {
        int y0, y1; float ly0, ly1;
        orc_src_index_r(Y, S, H, &y0, &y1, &ly0, &ly1);
        for (int X = 0; X < W; X++) {
            int x0, x1; float lx0, lx1;
            orc_src_index_r(X, S, W, &x0, &x1, &lx0, &lx1);
            float best = 0.0f; int bi = 0;
            for (int c = 0; c < C; c++) {
                float v = cls_label[c] * orc_bilerp(cam + c * ss, S, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
                if (c == 0 || v > best) { best = v; bi = c; }
            }
            lab_cam[(size_t)Y * W + X] = best <= bkg_thre ? 0 : (uint8_t)(bi + 1);
            float bp = 0.0f, bv = 0.0f; int ip = 0, iv = 0;
            for (int c = 0; c <= C; c++) {
                float v = orc_bilerp(seg + c * ss, S, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
                if (c == 0 || v > bp) { bp = v; ip = c; }
                float vv = (c == 0 || cls_label[c - 1] != 0.0f) ? v : -1e5f;        /* seg_validation :588 */
                if (c == 0 || vv > bv) { bv = vv; iv = c; }
            }
            lab_ps[(size_t)Y * W + X] = (uint8_t)ip;
            lab_vd[(size_t)Y * W + X] = (uint8_t)iv;
        }
    }
}

/* _fast_hist (evaluation.py:10-16) accumulated into hist[nc*nc] (row = truth).  pseudo != 0: the relabelling of
 * pseudo_scores (:43-46): truth := 255 where the prediction is 255 (those pixels drop out), prediction 255 -> 0. */
void orc_confusion(const uint8_t *gt, const uint8_t *pred, size_t n, int nc, int pseudo, int64_t *hist)
{
    for (size_t i = 0; i < n; i++) {
        int t = gt[i], p = pred[i];
        if (pseudo && p == 255) continue;
        if (t < nc) hist[(size_t)nc * t + p] += 1;
    }
}
