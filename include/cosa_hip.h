/*
 * cosa_hip.h -- C ABI of libcosa_hip.so: the MI355X (gfx950) implementation of CoSA's
 * per-iteration training hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every *_dev / device entry point takes DEVICE pointers and a `void *stream`
 *     (a hipStream_t; NULL = the null stream), enqueues work and returns without
 *     synchronising.  Return value: COSA_OK or a COSA_E* code; cosa_last_error()
 *     gives the message for the calling thread.
 *   - tensors are dense row-major ("NCHW") float32 unless a parameter says otherwise.
 *   - workspaces are caller-allocated; *_workspace_bytes() tells the size.  Nothing in the
 *     launch path calls hipMalloc/hipFree/hipDeviceSynchronize (graph-capturable).
 *   - reference file:line citations are relative to the CoSA repository root.
 */
#ifndef COSA_HIP_H
#define COSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COSA_OK 0
#define COSA_EINVAL 1   /* bad argument / unsupported shape            */
#define COSA_EHIP 2     /* a HIP runtime call failed                    */
#define COSA_ERANGE 3   /* lattice key left the packable range          */
#define COSA_ENOMEM 4   /* workspace too small                          */

int cosa_abi_version(void);
const char *cosa_last_error(void);

/* ---------------------------------------------------------------------------------------
 * utils/torch_helper.py:354-367  denormalize_img
 *   out = float(uint8(img*std+mean)) / 255      img,out [B,3,H,W]
 * ------------------------------------------------------------------------------------- */
int cosa_denormalize_img(const float *img, float *out, int B, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:264-270  per-(b,c) plane  x -= min(x); x /= max(x) + 1e-5  (in place)
 *   cam [BC, HW]; `active` (optional, [BC]): planes whose entry is 0 are known all-zero and skipped
 * ------------------------------------------------------------------------------------- */
int cosa_cam_minmax_norm(float *cam, int BC, int HW, const float *active /* [BC] or NULL */, void *stream);
/* the same result bit for bit with a plane spread over several workgroups; workspace: 2 * BC unsigned ints of device memory */
int cosa_cam_minmax_norm_ws(float *cam, int BC, int HW, const float *active /* [BC] or NULL */, void *workspace, void *stream);

/* utils/seg_helper.py:247-250: F.interpolate(imgs, size, mode="bilinear", align_corners=False) of the teacher's input batch to the 0.5x / 1.5x
 * scales; src [planes, H, W] fp32 -> dst [planes, OH, OW] (ATen's formula incl. its fused source-index multiply-add: within 1 ulp of the image range of the CPU operator) */
int cosa_resize_bilinear(const float *src, float *dst, int planes, int H, int W, int OH, int OW, void *stream);
/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:252-270  fused tail of multi_scale_camseg for ONE scale:
 *   up   = bilinear(src[2b,C,h,w] -> (S,S), align_corners=False)
 *   v    = mode 0: relu(max(up[:b], flip_w(up[b:])))          (cam / cam_aux, :253-258)
 *          mode 1: up[:b] + flip_w(up[b:])                    (seg,           :260-262)
 *   dst  = accumulate ? dst + v : v                           (sum over scales, :264,273)
 *   src [2*B, C, h, w]   dst [B, C, S, S]
 *   active (optional, the image-level labels): channels with active[b,c]==0 are written as 0 --
 *   what cam_validation (:547-551) makes of them one call later -- and cost nothing.
 * ------------------------------------------------------------------------------------- */
int cosa_cam_flip_merge_upsample(const float *src, float *dst, int B, int C, int h, int w, int S,
                                 int mode, int accumulate, const float *active /* [B,C] or NULL */, void *stream);
/* the same for a destination buffer that the previous call filled under the activity map prev_active [B*C]: planes absent then and now are
 * not touched (they are still zero) -- the training loop keeps its CAM buffers from step to step                                        */
int cosa_cam_flip_merge_upsample_reuse(const float *src, float *dst, int B, int C, int h, int w, int S, int mode, int accumulate,
                                       const float *active, const float *prev_active, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:721-797  cam2mask (+ _refine_cams), with cam_validation (:547-551)
 * folded in and, optionally, models/PAR.py:64-91 as the refine_model.
 *
 *   images  [B,3,S,S]  de-normalised image in [0,1] (read only when par_iters > 0)
 *   boxes   [B,4] int32 device (h0,h1,w0,w1)
 *   cams    [B,C,S,S]  cams; fold_validation=1: raw cams, multiplied by labels here
 *                      (cam_validation); 0: the caller already did (reference call order)
 *   labels  [B,C]      {0,1}
 *   mask    [B,S,S]    float32 out: {0..C, ignore_index}
 *   downscale 2 or 0;  par_iters 0 => refine_model=None;  dilations host int[n_dil]
 * ------------------------------------------------------------------------------------- */
size_t cosa_cam2mask_workspace_bytes(int B, int C, int S, int downscale, int n_dil);
int cosa_cam2mask(const float *images, const int32_t *boxes, const float *cams, const float *labels,
                  float *mask, int B, int C, int S, float thr_hi, float thr_lo, int downscale,
                  int fold_validation, const int *dilations, int n_dil, int par_iters, float ignore_index,
                  void *workspace, size_t workspace_bytes, void *stream);

/* The same for G CAM sets of the SAME images in one pass (the training step's two calls, main.py:137-166: main CAMs
 * and auxiliary CAMs with their own thresholds).  cams / masks: host arrays of G device pointers ([B,C,S,S] / [B,S,S]),
 * thr_hi / thr_lo: host float[G].  Bit-identical to G cosa_cam2mask calls; the affinities of the refine model are built
 * once and streamed once per propagation step for all sets.  thr_dev (optional): device float[G][2] = (hi, lo) per set,
 * read by the kernels instead of the host values -- the adaptive thresholds of main.py:138-151 without a host sync. */
size_t cosa_cam2mask_multi_workspace_bytes(int G, int B, int C, int S, int downscale, int n_dil);
int cosa_cam2mask_multi(const float *images, const int32_t *boxes, const float *const *cams, const float *labels,
                        float *const *masks, const float *thr_hi, const float *thr_lo, const float *thr_dev, int G, int B, int C,
                        int S,
                        int downscale, int fold_validation, const int *dilations, int n_dil, int par_iters,
                        float ignore_index, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * dataloaders/voc.py:262-275 `__transforms` (transforms.py:10-28,52-77,104-120,150-202; randaug.py:58-130) for a batch:
 * random_scaling -> random_fliplr -> random_crop -> GaussianBlur -> weak / OneOf(9 ops) strong -> ToTensor + Normalize,
 * with the random draws made by the caller (cosa_amd/dataloaders/augment.py: draw_params) and shipped as records.
 *   raw      device uint8: the decoded images, HWC, packed (record.raw_off = byte offset)
 *   records  device, B records of cosa_augment_record_bytes() bytes (26 int32, layout in csrc/aug_kernels.hip: AugImage)
 *   wimg, simg   device float32 [B,3,S,S] out;   crop_u8 / weak_u8 / strong_u8: optional device uint8 [B,S,S,3] stages
 * Bit-exact with Pillow's integer / fixed-point arithmetic (resize BILINEAR, GaussianBlur, ImageOps, ImageEnhance).
 * ------------------------------------------------------------------------------------- */
int cosa_augment_record_bytes(void);
size_t cosa_augment_workspace_bytes(int B, int S, int max_rows);
int cosa_augment_batch(const uint8_t *raw, const void *records, int B, int S, int max_rows, float *wimg, float *simg,
                       uint8_t *crop_u8, uint8_t *weak_u8, uint8_t *strong_u8, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:924-943  rungmm -- the adaptive-threshold fit of main.py:138-151,174-184:
 * scikit-learn GaussianMixture(modal, weights 1/modal, means (min, median, max) | (min, max), precisions 1,
 * tol, reg_covar, max_iter).fit_predict on the queue samples above the filter threshold, then
 * max(samples of component 0) [and min(samples of component 2) for modal = 3].  One launch, float64, no host sync.
 *   sorted  device float64: the samples above the filter threshold in ascending order (positive)
 *   n_dev   device int64: how many of them (<= capacity)
 *   out     device float64[13]: [0] low threshold, [1] high threshold (NaN for modal 2), [2] EM iterations,
 *           [3] status bits (1: component 0 empty, 2: component 2 empty, 4: fewer samples than components,
 *           8: grid barrier expired), [4..] means, [7..] weights, [10..] 1/sigma
 * ------------------------------------------------------------------------------------- */
size_t cosa_gmm_workspace_bytes(void);
int cosa_gmm_fit_thresholds(const double *sorted, const long long *n_dev, long long capacity, int modal, double tol,
                            double reg_covar, int max_iter, double *out, void *workspace, size_t workspace_bytes,
                            void *stream);

/* ---------------------------------------------------------------------------------------
 * models/PAR.py:64-91  PAR.forward for a batch of same-sized images / mask stacks.
 *   imgs [B,3,h,w]   masks [B,K,h,w] (in)   out [B,K,h,w]
 * ------------------------------------------------------------------------------------- */
size_t cosa_par_workspace_bytes(int B, int K, int h, int w, int n_dil);
int cosa_par_forward(const float *imgs, const float *masks, float *out, int B, int K, int h, int w,
                     const int *dilations, int n_dil, int num_iter,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/bilateralfilter/bilateralfilter.hpp:10-12 (SWIG module `bilateralfilter`,
 * bilateralfilter.i:21-25).  HOST-pointer drop-ins with the reference's exact argument list
 * (the SWIG typemaps turn each (ptr,len) pair into one NumPy array); they stage through the
 * GPU and synchronise.  `out(s)` is written in place.
 * ------------------------------------------------------------------------------------- */
void bilateralfilter(float *image, int len_image, float *in, int len_in, float *out, int len_out,
                     int H, int W, float sigmargb, float sigmaxy);
void bilateralfilter_batch(float *images, int len_images, float *ins, int len_ins, float *outs, int len_outs,
                           int N, int K, int H, int W, float sigmargb, float sigmaxy);

/* Device-resident form of the same filter (utils/bilateralfilter/permutohedral.cpp:115-297
 * init, :507-571 compute): images [N,3,H,W] (0..255), in/out [N,K,H,W].                     */
size_t cosa_bilateral_workspace_bytes(int N, int K, int H, int W);
int cosa_bilateralfilter_batch_dev(const float *images, const float *ins, float *outs,
                                   int N, int K, int H, int W, float sigmargb, float sigmaxy,
                                   int32_t *lattice_sizes /* [N] device, may be NULL */,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:864-903  DenseEnergyLossFunction, device resident.
 *   forward:  Gate = clamp(ROI - max_k seg, 0); Gate[unlabel] = 1; segm = seg*ROI;
 *             AS = BF(images, segm) * Gate;  loss = -<segm, AS> / N
 *   backward: grad_seg = -2 * grad_out * AS / N * ROI
 *   images [N,3,H,W] 0..255, seg [N,K,H,W], roi [N,H,W], unlabel [N,H,W] uint8
 *   AS [N,K,H,W] out (kept for backward), loss: 1 float (device)
 * ------------------------------------------------------------------------------------- */
int cosa_dense_energy_forward(const float *images, const float *seg, const float *roi, const uint8_t *unlabel,
                              float *AS, float *loss, int N, int K, int H, int W,
                              float sigmargb, float sigmaxy,
                              void *workspace, size_t workspace_bytes, void *stream);
/* the forward in two halves, so that the image-only half overlaps the networks: `prepare` takes the NORMALISED strong image
 * [N,3,S,S], writes F.interpolate(denormalize_img(.), scale_factor=0.5) to s_img [N,3,S/2,S/2] and builds the lattice in
 * `workspace`; `forward_prepared` (H = W = S/2) runs the filter through it.  Same K and workspace for both calls.          */
int cosa_dense_energy_prepare(const float *simg, float *s_img, int N, int K, int S, float sigmargb, float sigmaxy,
                              void *workspace, size_t workspace_bytes, void *stream);
int cosa_dense_energy_forward_prepared(const float *seg, const float *roi, const uint8_t *unlabel, float *AS, float *loss,
                                       int N, int K, int H, int W, float sigmargb, float sigmaxy,
                                       void *workspace, size_t workspace_bytes, void *stream);
int cosa_dense_energy_backward(const float *AS, const float *roi, const float *grad_out /* 1 float, device */,
                               float *grad_seg, int N, int K, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------
 * models/vit/vit.py:119-137  Attention core: softmax(q k^T * scale) v per head, fused (the
 * [B,H,N,N] score tensor is never materialised).  bf16 in/out, fp32 softmax and accumulation.
 *   qkv [B,N,3,H,64] bf16 (output of the qkv projection)   out [B,N,H*64] bf16
 *   lse [B,H,N] f32 (log-sum-exp of the scaled scores, kept for the backward pass)
 *   V is read in place (its V^T fragments come from the transposing LDS read): the workspace is a token 256 bytes and
 *   cosa_attn_prepare_vt a no-op, both kept for callers written against the earlier V^T-copy version; `flags` bits 8 / 9
 *   force 4 / 2 waves per workgroup (default: by the number of rounds, attn_kernels.hip); bit 10 marks a pass WITHOUT backward (the
 *   teacher's / evaluation's torch.no_grad() calls of the same module): the kernel may then hold q pre-multiplied by scale * log2(e) in
 *   the operand type and feed the softmax's running maximum through the score MFMAs (one more rounding of q; lse is the log of the sum of
 *   the ROUNDED probabilities); the other bits are ignored.
 * ------------------------------------------------------------------------------------- */
size_t cosa_attn_workspace_bytes(int B, int N, int H);
int cosa_attn_prepare_vt(const void *qkv, int B, int N, int H, void *workspace, size_t workspace_bytes, void *stream);
int cosa_attn_fwd(const void *qkv, void *out, float *lse, int B, int N, int H, int head_dim, float scale,
                  int flags, uint64_t *stamps /* NULL, or 64 x {min start, max end} 100 MHz device ticks of this launch (workgroup id & 63 picks the pair) */,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the same attention (autograd of vit.py:128-134): dqkv [B,N,3,H,64] bf16 from
 * dout [B,N,H*64]; P is recomputed from lse, nothing of size N^2 touches HBM; no atomics
 * (one kernel owns query blocks for dQ, one owns key blocks for dK/dV); K^T / Q^T / dO^T operands come
 * from transposing LDS reads of the row-major tiles (no transposed copies): the workspace only holds
 * delta = rowsum(dO * O), [B,H,N] fp32.                                                        */
size_t cosa_attn_bwd_workspace_bytes(int B, int N, int H);
int cosa_attn_bwd(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                  int B, int N, int H, int head_dim, float scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * models/vit/vit.py:96-102,119-137,154-158  the nn.Linear projections of a ViT block with their
 * element-wise tails fused:   Y = X[M,K] * W[N,K]^T + bias[N]   (bf16 operands, fp32 accumulate)
 *   epilogue 0: Y bf16                      (attn.qkv)
 *   epilogue 1: Y bf16 = gelu_erf(.)        (mlp.fc1 + nn.GELU)
 *   epilogue 2: Y fp32 = residual fp32 + .  (attn.proj / mlp.fc2 + the block's residual add;
 *                                            Y may alias residual)
 *   N % 128 == 0, K % 64 == 0.
 * cosa_layernorm: nn.LayerNorm(768, eps) over the fp32 residual stream -> bf16 (and/or fp32).
 * ------------------------------------------------------------------------------------- */
int cosa_gemm_bf16(const void *X, const void *W, const void *bias, const float *residual, void *Y,
                   int M, int N, int K, int epilogue, void *stream);
/* weight (and bias) gradient of the same Linear: dW[N,K] (fp32) = (zero_first ? 0 : dW) + dY[M,N]^T X[M,K]  (bf16 operands);
 * db[N] (fp32, optional) = (zero_first ? 0 : db) + column sums of dY.  N % 128 == 0, K % 128 == 0.  Deterministic: the token range is
 * split over workgroups, every split writes an fp32 partial slab into `workspace` (cosa_gemm_wgrad_workspace_bytes) and a second
 * kernel adds the slabs in split order -- no float atomics, the same bits every run.                                           */
size_t cosa_gemm_wgrad_workspace_bytes(int M, int N, int K);
int cosa_gemm_wgrad_bf16(const void *dY, const void *X, float *dW, float *db, int M, int N, int K, int zero_first,
                         void *workspace, size_t workspace_bytes, void *stream);
/* The same weight / bias gradients for MANY linears that share M, in one persistent launch (the student's encoder: autograd of
 * models/vit/vit.py:96-137 for all twelve blocks, launched when the backward pass has produced every dY).  With hundreds of 256 x 128 tiles
 * no item needs split-K: every tile runs the whole token loop and writes dW (and db) once -- overwritten, never accumulated; no workspace,
 * no atomics, the same bits every run.  `items`: host array, read during the call.                                                        */
typedef struct CosaWgradItem { const void *dY, *X; float *dW, *db; int N, K; } CosaWgradItem;
int cosa_gemm_wgrad_batched(const CosaWgradItem *items, int n_items, int M, void *stream);
/* models/decoder/conv_head.py:11-41  LargeFOV's 3x3 dilated, bias-free convolution on NHWC tokens (implicit GEMM, optional ReLU):
 *   X: image b = rows [b*img_rows + row_off, +h*w) of a [*, ldx] bf16 matrix (so the token tensor minus its cls row needs no copy)
 *   Wt [9][Cout][Cin] bf16 (tap-major: t = ky*3 + kx);  Y [B*h*w, Cout] bf16;  padding = dilation                               */
int cosa_conv3x3_dilated_nhwc(const void *X, const void *Wt, void *Y, int B, int h, int w, int Cin, int Cout, int dilation,
                              int img_rows, int row_off, int ldx, int relu, void *stream);
/* its weight gradient (autograd of the same conv): dW9 [Cout][9*Cin] fp32, tap-major columns (t*Cin + c), = (zero_first ? 0 : dW9) +
 * dY[B*h*w, Cout]^T im2col(X), the im2col implicit in the operand addressing; X is addressed as in the forward call.
 * (The input gradient is the forward entry point itself on dY with Wt'[t][c][o] = Wt[8-t][o][c].)                              */
int cosa_conv3x3_dilated_wgrad(const void *dY, const void *X, float *dW9, int B, int h, int w, int Cin, int Cout, int dilation,
                               int img_rows, int row_off, int ldx, int zero_first, void *workspace, size_t workspace_bytes,
                               void *stream);      /* workspace: cosa_gemm_wgrad_workspace_bytes(B*h*w, Cout, 9*Cin) */
void cosa_gemm_set_variant(int v);   /* 0 (default): per-shape choice; 1: the 128x128 two-stage kernel; 6: the persistent 256x256 kernel;
                                      * 9: the same on 256x192 jobs; 7 / 8 / 61-65: store-policy and timing variants of it (tools/) */
/* persistent-grid policy of cosa_gemm_bf16: 0 (default) one workgroup per CU; 1 the workgroups balanced over the rounds the launch needs
 * anyway (600 jobs: 200 workgroups x 3 instead of 256 x 2.3; launches of fewer than 40 000 rows, i.e. the student's), which leaves CUs to
 * concurrently running kernels -- RCCL's channels when the
 * process is one rank of a data-parallel job (utils/misc.py:439 init_distributed_mode + main.py:49-50 DDP)                              */
void cosa_gemm_set_grid_policy(int balanced);
void cosa_gemm_set_grid_policy_f16(int balanced);
/* measurement hook: the NEXT cosa_gemm_bf16 launch (persistent 256x256 kernel only) writes its device-clock span
 * (100 MHz s_memrealtime: min start / max end over workgroups) to 64 pairs slot[2s], slot[2s + 1] (uint64, s = workgroup id & 63,
 * caller-initialised to max / 0: the span is the min over the starts and the max over the ends).
 * One-shot; used by bench.py because HIP events cannot be recorded inside a captured hipGraph.                       */
void cosa_gemm_set_stamp_slot(void *slot);
/* vit.py:254-262 (PatchEmbed: stride-16 conv) as a GEMM needs the image as im2col rows; the teacher's passes run every scale as
 * cat(x, x.flip(-1)) (seg_helper.py:241-246).  cols [flips*B*(H/P)*(W/P), C*P*P] in a 16-bit type (dtype 1 = bf16, 2 = fp16) from
 * x [B,C,H,W] fp32: rows of the images first, then (flips = 2) of their horizontal mirror images.  P % 8 == 0, P | H, P | W.
 * dtype 3: the rows as fp16c8 operand rows (hi fp16 | lo8 | hi8 | aug = (1, 1, 0, ...), 4*C*P*P + 128 bytes each; see cosa_c8_rows).    */
int cosa_im2col_flip(const float *x, void *cols, int B, int C, int H, int W, int P, int flips, int dtype, void *stream);
/* the dtype-3 form writing into a token matrix with cls_rows free rows in front of every image's patch rows (kept zero by the caller): the
 * patch projection's fp32 residual epilogue then produces the residual stream [images, 1 + n, D] in place, without concatenations */
int cosa_im2col_flip_c8_tokens(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream);
/* the same token-shaped operand as split rows (hi | lo | aug = (1, 1, 0, ...); bf16 halves, `_f16`: fp16 halves) for the three-term (bf16x3 /
 * fp16x3) patch projection: rows [flips * B * (h*w + cls_rows), 2 C P P + 64], class-token rows untouched (vit.py:254-262, seg_helper.py:241-246) */
int cosa_im2col_flip_split_tokens(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream);
int cosa_im2col_flip_split_tokens_f16(const float *x, void *rows, int B, int C, int H, int W, int P, int flips, int cls_rows, void *stream);
/* vit.py:303-313 (prepare_tokens: cat(cls_token, patch tokens) + interpolated pos_embed) for the no-grad passes, written straight into
 * the fp32 residual stream: out [B, n+1, D] = (cls [D] | tok [B, n, D]) + pos [n+1, D]; tok / cls / pos share one 16-bit type
 * (dtype 1 = bf16, 2 = fp16), each sum is rounded to that type before it is widened (the 16-bit torch expression's value); D % 8 == 0. */
int cosa_embed_finish(const void *tok, const void *cls, const void *pos, float *out, int B, int n, int D, int dtype, void *stream);
/* models/__init__.py:190-192 (classifier / aux_classifier as 1x1 convs over the tokens) and conv_head.py:38 (conv8): the narrow heads
 * Y[M, N] (fp32, columns [col0, col0+N) of rows with stride ldy; N of a few dozen rows, slices of 32 inside the kernel) = X W^T, fp32
 * accumulation, fixed reduction order per row
 * (results do not depend on what else is in the batch).  X: image b = rows_per_img rows at X + b*img_stride (elements), row stride
 * ldx; dtype 0: fp32 X and W, 1: bf16, 2: fp16 X and W (round_bf16 = 1 rounds the result to the operand precision).                               */
int cosa_head_gemm(const void *X, const void *W, float *Y, int M, int N, int K, int rows_per_img, long long img_stride,
                   int ldx, int dtype, int round_bf16, int ldy, int col0, void *stream);
/* autograd of the same heads (the reference trains them through torch autograd: models/__init__.py:190-204, conv_head.py:38).
 * dX [M,K] bf16 = dY [M,N] (fp32, contiguous) W [N,K] (bf16);  dW [N,K] fp32 = dY^T X (X [M,K] bf16, contiguous), summed in a fixed
 * order (workspace: cosa_head_gemm_wgrad_workspace(M, K) bytes of device memory).  K % 128 == 0; the weight gradient takes N <= 32
 * rows per call (wider heads in slices).                                                                                          */
int cosa_head_gemm_dgrad(const float *dY, const void *W, void *dX, int M, int N, int K, void *stream);
size_t cosa_head_gemm_wgrad_workspace(int M, int K);
int cosa_head_gemm_wgrad(const float *dY, const void *X, float *dW, void *workspace, int M, int N, int K, void *stream);
int cosa_layernorm(const float *x, const void *gamma, const void *beta, void *y_bf16, float *y_f32,
                   int rows, int dim, float eps, void *stream);
/* the student's training path (autograd of models/vit/vit.py:96-102,119-137) on the same GEMM kernels:
 *   cosa_gemm_bf16_dual_gelu    mlp.fc1 forward: H = X W^T + b (bf16, kept for GELU') and A = gelu_erf(H) (bf16) from one pass
 *   cosa_gelu_backward          dH = dA * gelu_erf'(H) (bf16, n % 8 == 0)
 *   cosa_transpose_cast_batched bf16 W^T shadows of the fp32 master weights (one launch for all tensors): the input gradient
 *                               dX = dY W is then cosa_gemm_bf16(dY, W^T, zeros) -- the forward kernel, no second GEMM family   */
/* dst [B][n] fp32 = src [n] for every b (n % 4 == 0): the fp32 token stream of a no-grad pass starts as (cls + pos_0 | pos rows) per image,
 * models/vit/vit.py:283-300; the patch projection then adds into it in place */
int cosa_broadcast_rows(const float *src, float *dst, int B, long long n, void *stream);
/* vit.py:288-291: bicubic resize of the frozen 14 x 14 position grid to the token grid as a 16-tap gather (idx / wgt [P,16]: the non-zero
 * entries of the interpolation matrix's rows): out[P, D] fp32 = sum_t wgt[p][t] * pe[idx[p][t]][:]                                 */
int cosa_pos_resize(const float *pe, const int *idx, const float *wgt, float *out, int P, int D, void *stream);
int cosa_gemm_bf16_dual_gelu(const void *X, const void *W, const void *bias, void *H, void *A, int M, int N, int K, void *stream);
int cosa_gelu_backward(const void *dA, const void *H, void *dH, long long n, void *stream);
size_t cosa_transpose_record_bytes(void);
int cosa_transpose_cast_batched(const void *records, int n, int total_tiles, void *stream);

/* The same kernels with IEEE fp16 operands (fp32 accumulation, same MFMA rate; gemm_kernels.hip / attn_kernels.hip built a second
 * time with -DCOSA_OP_F16=1): the no-grad passes -- the teacher's six multi-scale forwards per step (utils/seg_helper.py:232-275) and
 * evaluation -- may run on fp16 operands, which keeps the pseudo-label maps within the stated tolerance of the fp32 reference where
 * bf16's 8 significant bits do not.  Argument lists are those of the bf16 entry points, every "bf16" buffer
 * being fp16 instead.                                                                                                            */
int cosa_gemm_f16(const void *X, const void *W, const void *bias, const float *residual, void *Y,
                  int M, int N, int K, int epilogue, void *stream);
int cosa_gemm_wgrad_f16(const void *dY, const void *X, float *dW, float *db, int M, int N, int K, int zero_first,
                        void *workspace, size_t workspace_bytes, void *stream);
int cosa_layernorm_f16(const float *x, const void *gamma, const void *beta, void *y_f16, float *y_f32,
                       int rows, int dim, float eps, void *stream);
int cosa_conv3x3_dilated_nhwc_f16(const void *X, const void *Wt, void *Y, int B, int h, int w, int Cin, int Cout, int dilation,
                                  int img_rows, int row_off, int ldx, int relu, void *stream);
int cosa_conv3x3_dilated_wgrad_f16(const void *dY, const void *X, float *dW9, int B, int h, int w, int Cin, int Cout, int dilation,
                                   int img_rows, int row_off, int ldx, int zero_first, void *workspace, size_t workspace_bytes,
                                   void *stream);
void cosa_gemm_set_variant_f16(int v);
void cosa_gemm_set_stamp_slot_f16(void *slot);
size_t cosa_attn_workspace_bytes_f16(int B, int N, int H);
int cosa_attn_prepare_vt_f16(const void *qkv, int B, int N, int H, void *workspace, size_t workspace_bytes, void *stream);
int cosa_attn_fwd_f16(const void *qkv, void *out, float *lse, int B, int N, int H, int head_dim, float scale,
                      int flags, uint64_t *stamps, void *workspace, size_t workspace_bytes, void *stream);
size_t cosa_attn_bwd_workspace_bytes_f16(int B, int N, int H);
int cosa_attn_bwd_f16(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                      int B, int N, int H, int head_dim, float scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * bf16x3 ("split") operands: the parity-grade precision of the no-grad passes (teacher pseudo-labels, utils/seg_helper.py:232-275;
 * evaluation).  The reference runs fp32; bf16's 8 significant bits put the normalised CAMs ~1e-2 and the label maps
 * ~0.2 % away from it at 448^2, so every MFMA operand of these passes can instead be carried as hi = bf16(v), lo = bf16(v - hi)
 * (16 significant bits) with three MFMA terms per product (hi*hi + lo*hi + hi*lo) and fp32 accumulation -- still bf16 MFMA, 3x the
 * work.  A split row of logical width K is [hi (K) | lo (K) | aug (64)], row stride 2K + 64; aug = (1, 1, 0, ...) for activations and
 * (bias_hi, bias_lo, 0, ...) for weight row n, so nn.Linear's bias rides in the GEMM as one more K tile.
 *   cosa_split_rows       src fp32 [R, K] (row stride src_ld) (+ bias fp32 [R] | ones) -> dst split rows [R, 2K + 64]
 *   cosa_layernorm_split  nn.LayerNorm(768, eps), fp32 gamma / beta, over the fp32 residual stream -> split rows and/or fp32
 *   cosa_gemm_bf16x3      Y = Xs Ws^T (bias inside Ws): epilogue 0 / 1 (GELU): Y bf16 [M, ldy >= 2N] = [hi | lo]; epilogue 2: Y fp32
 *                         [M, N] = residual + . (may alias);  zeros: N bf16 zeros (the kernels' bias operand)
 *   cosa_attn_fwd_bf16x3  attention on the split qkv rows [B*N, ldq] = [hi (3*H*64) | lo ...] -> split rows [B*N, ldo >= 2*H*64 + 64]
 *                         for the output projection; lse optional
 * ------------------------------------------------------------------------------------- */
int cosa_split_rows(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream);
int cosa_layernorm_split(const float *x, const float *gamma, const float *beta, void *y_split, float *y_f32, int rows, int dim,
                         float eps, void *stream);
int cosa_gemm_bf16x3(const void *Xs, const void *Ws, const void *zeros, const float *residual, void *Y,
                     int M, int N, int K, int epilogue, int ldy, void *stream);
int cosa_attn_fwd_bf16x3(const void *qkv_split, void *out_split, float *lse, int B, int N, int H, int head_dim, float scale,
                         int ldq, int ldo, uint64_t *stamps /* optional device-clock span of the launch, as cosa_attn_fwd */, void *stream);
/* fp16x3 (round 6): the same four entry points with hi = fp16(v), lo = fp16(v - hi) halves -- 11 + 11 significant bits (a lo half below 2^-14
 * is an fp16 subnormal: an absolute 2^-24), three fp16 MFMA terms, same cost as bf16x3 and ~10x closer to the fp32 reference
 * (tools/sim_precision_map.py scheme `hh`).  Same layouts with fp16 halves; the attention carries its probabilities scaled by 2^10 so that
 * their lo halves stay normal numbers (the factor cancels in O / l).  Teacher mode string: "fp16x3". */
int cosa_split_rows_f16(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream);
int cosa_layernorm_split_f16(const float *x, const float *gamma, const float *beta, void *y_split, float *y_f32, int rows, int dim,
                             float eps, void *stream);
int cosa_gemm_f16x3(const void *Xs, const void *Ws, const void *zeros, const float *residual, void *Y,
                    int M, int N, int K, int epilogue, int ldy, void *stream);
int cosa_attn_fwd_f16x3(const void *qkv_split, void *out_split, float *lse, int B, int N, int H, int head_dim, float scale,
                        int ldq, int ldo, uint64_t *stamps, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/seg_helper.py:961-996 (DenseCRF / crf_inference_infv2, final evaluation only): the position-only Gaussian kernel of the dense
 * CRF -- pydensecrf's addPairwiseGaussian: features (x / sxy, y / sxy) -- as a permutohedral-lattice filter on a 2-D lattice (the lattice
 * kernels of the bilateral filter, csrc/permuto_kernels.hip, compiled a second time with -DCOSA_PD=2).  ins / outs [N, K, H, W] fp32.
 * The bilateral kernel of the CRF is cosa_bilateralfilter_batch_dev itself; the mean-field update around the two filters is host code
 * (cosa_amd/utils/seg_helper.py: DenseCRF).  pydensecrf is not under the reference tree: parity of this row is unpinned (oracle/crf_oracle.py).
 * ------------------------------------------------------------------------------------- */
/* the lattice's own stable LSD radix sort of (key, value) pairs by the low `bits` bits of the key (csrc/radix_sort.hpp; exported for the tests) */
size_t cosa_radix_sort_workspace_bytes(long long n);
int cosa_radix_sort_pairs(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, long long n, int bits,
                          void *workspace, size_t workspace_bytes, void *stream);
size_t cosa_lattice_filter_d2_workspace_bytes(int N, int K, int H, int W);
int cosa_lattice_filter_d2(const float *ins, float *outs, int N, int K, int H, int W, float sigmaxy, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * fp16c8 operands: the parity-grade precision of the no-grad passes at 2x (not 3x) the MFMA work of the 16-bit path
 * (teacher pseudo-labels, utils/seg_helper.py:232-275; replaces the fp32 arithmetic of models/vit/vit.py:96-137 on the no-grad path).
 * A value v is carried as hi = fp16(v) plus two e5m2 bytes: hi8 = e5m2(hi) and lo8 = e5m2((v - hi) * 2^11); a product is
 * x_hi w_hi (fp16 MFMA) + 2^-11 (x_lo8 w_hi8 + x_hi8 w_lo8) (block-scaled 8-bit MFMA, K = 128 per instruction, twice the fp16 rate; the
 * 2^-11 is the instruction's E8M0 scale).  e5m2 has fp16's exponent range: no per-tensor statistics or scales exist.  A c8 row of logical
 * width K is, in bytes, [hi fp16 (2K) | lo8 (K) | hi8 (K) | aug fp16 (128)], row stride 4K + 128 (= the bf16x3 stride); aug = (1, 1, 0, ...)
 * for activations and (bias_hi, bias_lo, 0, ...) for weight row n (the bias rides in the GEMM as one more fp16 K tile).
 *   cosa_c8_rows          src fp32 [R, K] (row stride src_ld) (+ bias fp32 [R] | ones) -> dst c8 rows; K % 128 == 0
 *   cosa_layernorm_c8     nn.LayerNorm(768, eps), fp32 gamma / beta, over the fp32 residual stream -> c8 rows and/or fp32
 *   cosa_gemm_f16c8       Y = Xs Ws^T (bias inside Ws); N % 256 == 0, K % 128 == 0.  epilogue 0: Y fp16 [M, ldy >= N] (plain: the
 *                         qkv projection feeds the fp16 attention kernel); 1 (GELU): Y = c8 rows [M, ldy = 2N + 64 fp16 units], hi | lo8 |
 *                         hi8 written, the augmentation block left to the caller; 2: Y fp32 [M, N] = residual + . (may alias);
 *                         zeros: N fp16 zeros (the kernel's unused bias operand)
 *   cosa_attn_fwd_f16c8   attention on plain fp16 qkv rows [B, N, 3, H, 64] -> c8 rows [B*N, 4*H*64 + 128 bytes] (incl. the augmentation
 *                         block) for the c8 output projection; lse optional
 * ------------------------------------------------------------------------------------- */
int cosa_c8_rows(const float *src, const float *bias, void *dst, int R, int K, long long src_ld, int ones, void *stream);
size_t cosa_c8_record_bytes(void);       /* { const float *src; const float *bias; void *dst; int rows, K, row0, qrows; }: the first qrows rows and bias
                                          * entries are multiplied by 64^-0.5 log2(e) first (the q third of a qkv projection, for cosa_attn_fwd_f16c8
                                          * called with scale = ln 2: the attention scale folded into the weights, no extra rounding of q) */
int cosa_c8_rows_batched(const void *records /* device */, int n_records, int total_rows, void *stream);   /* all weight matrices, one launch */
int cosa_layernorm_c8(const float *x, const float *gamma, const float *beta, void *y_c8, float *y_f32, int rows, int dim,
                      float eps, void *stream);
int cosa_gemm_f16c8(const void *Xs, const void *Ws, const void *zeros, const float *residual, void *Y,
                    int M, int N, int K, int epilogue, int ldy, void *stream);
int cosa_attn_fwd_f16c8(const void *qkv, void *out_c8, float *lse, int B, int N, int H, int head_dim, float scale,
                        uint64_t *stamps, void *stream);

/* ---------------------------------------------------------------------------------------
 * fp16c4 operands (round 4; csrc/c4.hpp): the fp16c8 scheme with FP4 (e2m1) correction terms in MX blocks -- both terms
 * 2^-11 (x_lo' w_hi + x_hi w_lo') in ONE stream of block-scaled MFMAs at 4x the fp16 rate (e5m2: 2x): ~1.58x instead of ~2.08x the MFMA work
 * of the plain 16-bit path at K = 768.  Same purpose and call sites as fp16c8 (the teacher's no-grad projections, models/vit/vit.py:96-137).
 * A c4 operand is a PAIR: rows [R][4K + 128 bytes] = [hi fp16 (2K) | 16-byte blocks (K) | unused (K) | aug fp16 (128)] -- the c8 stride; a
 * block holds the 16 lo' = (v - hi) 2^11 and the 16 hi copies of 16 consecutive features as e2m1 nibbles ([lo' | hi] for activations,
 * [hi | lo'] for weights) -- and a scale tensor of cosa_c4_scale_bytes(R, K) bytes with one E8M0 byte per block (smallest 2^e with
 * amax / 2^e <= 6; weights carry the 2^-11), laid out per 256-row panel and 128-feature tile as the GEMM's lanes read it (c4.hpp).  K % 256 == 0.
 *   cosa_c4_rows          src fp32 [R, K] (+ bias fp32 [R] | ones) -> rows + scales; weight != 0: weight block order / scale bias
 *   cosa_c4_rows_batched  the weight matrices of a network in one launch ({src, bias, dst, scales, rows, K, row0, qrows} records; qrows as for cosa_c8_rows_batched)
 *   cosa_layernorm_c4     nn.LayerNorm(768, eps), fp32 gamma / beta, over the fp32 residual stream -> c4 rows + scales and/or fp32
 *   cosa_gemm_f16c4       Y = Xs Ws^T (bias inside Ws); N % 256 == 0.  epilogue 0: Y fp16 [M, ldy >= N]; 1 (GELU): Y = c4 rows [M, ldy = 2N + 64
 *                         fp16 units] + Yscales (activation layout; the augmentation block left to the caller); 2: Y fp32 [M, N] = residual + .
 * ------------------------------------------------------------------------------------- */
size_t cosa_c4_scale_bytes(int rows, int K);
int cosa_c4_rows(const float *src, const float *bias, void *dst, void *scales, int R, int K, long long src_ld, int ones, int weight, void *stream);
size_t cosa_c4_record_bytes(void);       /* { const float *src; const float *bias; void *dst; void *scales; int rows, K, row0, qrows; } */
int cosa_c4_rows_batched(const void *records /* device */, int n_records, int total_rows, void *stream);
int cosa_layernorm_c4(const float *x, const float *gamma, const float *beta, void *y_c4, void *y_scales, float *y_f32, int rows, int dim,
                      float eps, void *stream);
int cosa_gemm_f16c4(const void *Xs, const void *Xscales, const void *Ws, const void *Wscales, const void *zeros, const float *residual,
                    void *Y, void *Yscales, int M, int N, int K, int epilogue, int ldy, void *stream);
/* attention on plain fp16 qkv rows [B, N, 3, H, 64] -> fp16c4 rows [B*N, 4*H*64 + 128 bytes] (incl. the augmentation block) for the c4 output
 * projection; out_scales = the scale tensor of the WHOLE operand these rows belong to, row0 = the index of out_c4's first row in it */
int cosa_attn_fwd_f16c4(const void *qkv, void *out_c4, void *out_scales, int row0, float *lse, int B, int N, int H, int head_dim, float scale,
                        uint64_t *stamps, void *stream);

/* ---------------------------------------------------------------------------------------
 * main.py:167-212 + utils/seg_helper.py:800-813,199-230  the student's dense losses, fused:
 *   seg_loss(main) and seg_loss(aux) of the bilinearly up-sampled logits, and the inputs of the
 *   dense-energy regulariser (softmax -> x0.5, ROI from boxes, nearest image / label), without
 *   materialising anything at [B,K,S,S].
 *   seg_lr [B,K,hs,ws] logits; maskA/maskB [B,S,S] labels {0..K-1,255}; simg [B,3,S,S] normalised image
 *   forward  -> sums[8] = {bgA_sum,bgA_cnt,fgA_sum,fgA_cnt,bgB_...}; s_seg [B,K,S/2,S/2]; s_img [B,3,S/2,S/2] (0..255);
 *               roi [B,S/2,S/2]; unlabel [B,S/2,S/2] u8   (then cosa_dense_energy_forward on these)
 *   backward -> grad_seg_lr [B,K,hs,ws] for  g_seg * (0.5*seg_loss_A + 0.5*seg_loss_B)  +  g_regw * energy
 *               (AS = the gated filter output kept by cosa_dense_energy_forward)
 *   Sums that meet from many threads (the eight loss sums, the gradient cells) are accumulated in 64-bit fixed point with INTEGER atomics
 *   in `workspace` (cosa_seg_loss_workspace_bytes) and converted at the end: order-independent, the same bits every run.
 * ------------------------------------------------------------------------------------- */
size_t cosa_seg_loss_workspace_bytes(int B, int K, int hs, int ws);
int cosa_seg_loss_forward(const float *seg_lr, const float *maskA, const float *maskB, const float *simg,
                          const int32_t *boxes, float *sums, float *s_seg, float *s_img, float *roi, uint8_t *unlabel,
                          int B, int K, int hs, int ws, int S, void *workspace, size_t workspace_bytes, void *stream);
int cosa_seg_loss_backward(const float *seg_lr, const float *maskA, const float *maskB, const float *sums, const float *AS,
                           const float *roi, const float *g_seg, const float *g_regw, float *grad_seg_lr,
                           int B, int K, int hs, int ws, int S, void *workspace, size_t workspace_bytes, void *stream);

/* F.multilabel_soft_margin_loss (main.py:127-128 on the classification logits; seg_helper.py:593-602 on relu(cam) against the resized teacher
 * probabilities) and its gradient in one pass: loss[0] = mean_r mean_c -(y log s(v) + (1-y) log s(-v)), v = relu ? max(x,0) : x;
 * grad = d loss / d x.  Element (r, c) of x / y / grad lies at (r / HW) * C * HW + c * HW + r % HW (HW = 1: row-major [R,C]; HW = h*w: NCHW).
 * workspace: ceil(R / 256) doubles.                                                                                                   */
int cosa_msm_loss(const float *x, const float *y, float *grad, float *loss, void *workspace, int R, int C, int HW, int relu, void *stream);
/* main.py:227-228 + utils/seg_helper.py:553-568,593-597: targets of cam_loss -- seg_refine_by_label(teacher seg, T) foreground
 * channels, bilinearly down-sampled to the CAM grid -- evaluated only at the pixels the down-sampling reads, straight from the
 * per-scale low-res teacher seg outputs seg_scales[i] [2B,K,hs[i],ws[i]] (original batch, then the flipped batch).
 *   out [B, K-1, oh, ow]                                                                                                      */
int cosa_cam_loss_targets(const float *const *seg_scales, const int *hs, const int *ws, int n_scales, const float *labels,
                          float *out, int B, int K, int S, int oh, int ow, float temperature, void *stream);
/* utils/seg_helper.py:210-230 (get_energy_loss: F.softmax over the classes of the full-resolution logits) + :199-203 (DenseEnergyLoss.forward:
 * bilinear resize of the probabilities by 0.5 = the mean of each 2x2 quad) in one pass, and the backward of the pair:
 *   logit [B,K,H,W] (H, W even)   out / grad_out [B,K,H/2,W/2]   grad_logit [B,K,H,W]                                                     */
int cosa_softmax_halfres_forward(const float *logit, float *out, int B, int K, int H, int W, void *stream);
int cosa_softmax_halfres_backward(const float *logit, const float *grad_out, float *grad_logit, int B, int K, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------
 * utils/torch_helper.py:261-293 (AdamW with the scheduled LR) + main.py:250-252 (teacher EMA) + the bf16 shadow
 * refresh, one multi-tensor pass.  `records`: device array, one per parameter tensor:
 *   { float *p; const float *g (NULL = frozen: EMA only); float *m, *v; float *teacher; bf16 *p16, *t16 (or NULL);
 *     float lr, wd; int64 n }   (cosa_optim_record_bytes() bytes each)
 * `chunks`: device array of {int tensor, int chunk} covering every tensor in pieces of cosa_optim_chunk_elems().
 * ------------------------------------------------------------------------------------- */
size_t cosa_optim_record_bytes(void);
int cosa_optim_chunk_elems(void);
int cosa_fused_adamw_ema(const void *records, const void *chunks, int n_chunks, float beta1, float beta2, float eps,
                         int step, float ema_momentum, void *stream);

/* ---------------------------------------------------------------------------------------
 * models/vit/vit.py:154-158  the student's pre-LN residual blocks (training, bf16 stream), element-wise side:
 *   cosa_add_layernorm_fwd: x_out = bf16(x + delta) (delta may be NULL: x_out optional), y = LayerNorm(x_out; gamma, beta, eps),
 *     mean / rstd [rows] kept for the backward.  All tensors bf16 [rows, 768] except mean / rstd (fp32).
 *   cosa_layernorm_bwd: dx = dLayerNorm(dy) + dskip (dskip optional: the gradient arriving at x_out through the skip path),
 *     dgamma / dbeta [768] fp32 (= or += with accumulate != 0), summed deterministically through `workspace`.
 * ------------------------------------------------------------------------------------- */
int cosa_add_layernorm_fwd(const void *x, const void *delta, const void *gamma, const void *beta, void *x_out, void *y,
                           float *mean, float *rstd, int rows, int dim, float eps, void *stream);
size_t cosa_layernorm_bwd_workspace_bytes(int rows, int dim);
int cosa_layernorm_bwd(const void *dy, const void *x_new, const float *mean, const float *rstd, const void *gamma,
                       const void *dskip, void *dx, float *dgamma, float *dbeta, int accumulate, int rows, int dim,
                       void *workspace, size_t workspace_bytes, void *stream);
/* fp32 residual stream (the student's default since round 4; the reference trains in fp32, main.py:124-246): the forward of a block's
 * LayerNorm is cosa_layernorm on the fp32 stream, the residual add is cosa_gemm_bf16's fp32 residual epilogue, and the backward is
 *   cosa_layernorm_bwd_f32: dx (fp32) = dLayerNorm(dy: bf16, or fp32 with dy_is_f32; x fp32, gamma bf16, eps) + dskip (fp32, optional), dx16 (optional) = bf16(dx)
 *     -- the dY operand of the preceding projection's gradient GEMMs; mean / rstd are recomputed from x; dgamma / dbeta as above. */
int cosa_layernorm_bwd_f32(const void *dy, int dy_is_f32, const float *x, const void *gamma, const float *dskip, float *dx, void *dx16,
                           float *dgamma, float *dbeta, int accumulate, int rows, int dim, float eps, void *workspace,
                           size_t workspace_bytes, void *stream);
/* models/__init__.py:163-206 + autograd: the gradient junction of the training path's heads.  g0, g1, g2: bf16 gradients [B, N - 1, dim] of the
 * consumers of the patch tokens (decoder, CAM head, pooled classification head; NULL = consumer absent); dx [B, N, dim] fp32 = their sum in
 * fp32 (order g0, g1, g2), zero in the class-token row.                                                                                   */
int cosa_token_junction_bwd(const void *g0, const void *g1, const void *g2, float *dx, int B, int N, int dim, void *stream);

/* ---------------------------------------------------------------------------------------
 * Evaluation path (SURVEY f-1; evaluation_engine.py:96-126,198-200, utils/seg_helper.py:515-546,581-591,
 * utils/evaluation.py:10-70).
 *   cosa_eval_labels: one launch replaces F.interpolate(cam), cam_to_label, F.interpolate(seg), seg_validation and the
 *     three argmaxes.  cam [B,C,S,S], seg [B,C+1,S,S] (either may be NULL), cls_label [B,C] -> uint8 maps [B,H,W]:
 *     lab_cam = argmax_c(cls*cam)+1 or 0 where the max <= bkg_thre; lab_ps = argmax(seg); lab_vd = argmax with the
 *     classes absent from cls_label masked to -1e5 (background always present).
 *   cosa_cam_to_label: cam_to_label on an already sized CAM [B,C,H,W] -> int64 label [B,H,W]; cls_label, boxes
 *     ([B,4] h0,h1,w0,w1; NULL = the reference's `img_box is None` return) and valid_cam (= cls*cam) optional.
 *   cosa_confusion_hist: hist[nc*t + p] += 1 over n pixels with truth t < nc (uint8 maps, 255 = ignore);
 *     pseudo != 0 drops the pixels whose prediction is 255 (pseudo_scores).  hist: nc*nc uint64, caller-zeroed, accumulates.
 * ------------------------------------------------------------------------------------- */
int cosa_eval_labels(const float *cam, const float *seg, const float *cls_label, int B, int C, int S, int H, int W,
                     float bkg_thre, uint8_t *lab_cam, uint8_t *lab_ps, uint8_t *lab_vd, void *stream);
int cosa_cam_to_label(const float *cam, const float *cls_label, int B, int C, int H, int W, float bkg_thre,
                      const int32_t *boxes, int ignore_mid, float high_thre, float low_thre, long long ignore_index,
                      long long *label, float *valid_cam, void *stream);
int cosa_confusion_hist(const uint8_t *gt, const uint8_t *pred, size_t n, int num_classes, int pseudo,
                        unsigned long long *hist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COSA_HIP_H */
