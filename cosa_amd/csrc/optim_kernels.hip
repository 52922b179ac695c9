// optim_kernels.hip -- one pass over the parameters per training step (gfx950, HBM-bound).
//
// Reference: utils/torch_helper.py:261-293 (PolyWarmupAdamW = torch AdamW with a scheduled LR), main.py:250-252 (EMA of
// the teacher, a Python loop over ~150 tensors).  Here AdamW, the EMA update and the refresh of the bf16 shadow copies that
// the forward passes read are ONE multi-tensor kernel: per parameter 20 B read (p, g, m, v, teacher) and 20 B written
// (p, m, v, teacher, 2 x bf16) instead of four separate sweeps (~72 B).  Frozen tensors (grad == NULL) only take the EMA.
#include "kernels.hpp"

namespace cosa {
namespace {

typedef __bf16 bf16;

struct TensorRec {          // one record per parameter tensor (device table)
    float *p;               // student master
    const float *g;         // gradient or NULL (frozen)
    float *m, *v;           // AdamW moments
    float *tp;              // teacher master
    bf16 *p16, *t16;        // bf16 shadows or NULL
    float lr, wd;           // per-group hyper-parameters (already scheduled)
    long long n;
    int t16_f16, p16_f16;   // the shadow's 16-bit type: 0 bf16, 1 IEEE fp16 (fp16-operand teacher passes)
};

struct ChunkRec { int tensor; int chunk; };

__device__ __forceinline__ uint2 pack16x4(const float (&v)[4], int f16)
{
    if (f16) {
        _Float16 o[4] = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return *reinterpret_cast<uint2 *>(o);
    }
    bf16 o[4] = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    return *reinterpret_cast<uint2 *>(o);
}
__device__ __forceinline__ void store16(bf16 *dst, float v, int f16)
{
    if (f16) *reinterpret_cast<_Float16 *>(dst) = (_Float16)v;
    else *dst = (bf16)v;
}
constexpr int kChunk = 65536;            // elements per block

__global__ __launch_bounds__(256) void adamw_ema_kernel(const TensorRec *__restrict__ recs, const ChunkRec *__restrict__ chunks,
                                                       float beta1, float beta2, float eps, float bc1, float bc2_sqrt, float ema)
{
    const ChunkRec c = chunks[blockIdx.x];
    const TensorRec t = recs[c.tensor];
    const long long base = (long long)c.chunk * kChunk;
    long long end = base + kChunk;
    end = end < t.n ? end : t.n;
    const float step_size = t.lr / bc1;
    const float decay = 1.0f - t.lr * t.wd;
    const bool vec = ((t.n & 3) == 0);
    if (vec) {
        for (long long i = base + threadIdx.x * 4; i < end; i += 1024) {
            float4 p = *reinterpret_cast<const float4 *>(t.p + i);
            float4 tp = *reinterpret_cast<const float4 *>(t.tp + i);
            float pv[4] = {p.x, p.y, p.z, p.w}, tv[4] = {tp.x, tp.y, tp.z, tp.w};
            if (t.g) {
                const float4 g = *reinterpret_cast<const float4 *>(t.g + i);
                float4 m = *reinterpret_cast<const float4 *>(t.m + i), v = *reinterpret_cast<const float4 *>(t.v + i);
                float gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    pv[e] *= decay;
                    mv[e] = beta1 * mv[e] + (1.0f - beta1) * gv[e];
                    vv[e] = beta2 * vv[e] + (1.0f - beta2) * gv[e] * gv[e];
                    const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
                    pv[e] -= step_size * (mv[e] / denom);
                }
                *reinterpret_cast<float4 *>(t.m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                *reinterpret_cast<float4 *>(t.v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                *reinterpret_cast<float4 *>(t.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) tv[e] = ema * tv[e] + (1.0f - ema) * pv[e];
            *reinterpret_cast<float4 *>(t.tp + i) = make_float4(tv[0], tv[1], tv[2], tv[3]);
            if (t.p16 && t.g) *reinterpret_cast<uint2 *>(t.p16 + i) = pack16x4(pv, t.p16_f16);
            if (t.t16) *reinterpret_cast<uint2 *>(t.t16 + i) = pack16x4(tv, t.t16_f16);
        }
    } else {
        for (long long i = base + threadIdx.x; i < end; i += 256) {
            float p = t.p[i], tp = t.tp[i];
            if (t.g) {
                const float g = t.g[i];
                p *= decay;
                const float m = beta1 * t.m[i] + (1.0f - beta1) * g;
                const float v = beta2 * t.v[i] + (1.0f - beta2) * g * g;
                p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
                t.m[i] = m; t.v[i] = v; t.p[i] = p;
                if (t.p16) store16(t.p16 + i, p, t.p16_f16);
            }
            tp = ema * tp + (1.0f - ema) * p;
            t.tp[i] = tp;
            if (t.t16) store16(t.t16 + i, tp, t.t16_f16);
        }
    }
}

}  // namespace
}  // namespace cosa

using namespace cosa;

extern "C" size_t cosa_optim_record_bytes(void) { return sizeof(TensorRec); }
extern "C" int cosa_optim_chunk_elems(void) { return kChunk; }

/* records: device array of n_tensors TensorRec (layout: 7 pointers, float lr, float wd, int64 n, int32 t16_f16, int32 p16_f16); chunks: device array of
 * n_chunks {int tensor, int chunk}.  step >= 1 is the AdamW step count used for bias correction.                       */
extern "C" int cosa_fused_adamw_ema(const void *records, const void *chunks, int n_chunks, float beta1, float beta2, float eps,
                                    int step, float ema_momentum, void *stream)
{
    COSA_REQUIRE(records && chunks && n_chunks > 0 && step >= 1, "cosa_fused_adamw_ema: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), static_cast<const TensorRec *>(records),
                       static_cast<const ChunkRec *>(chunks), beta1, beta2, eps, bc1, bc2_sqrt, ema_momentum);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
