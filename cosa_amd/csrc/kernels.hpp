// kernels.hpp -- internal launcher declarations shared between translation units.
#pragma once
#include "common.hpp"

namespace cosa {

// par_kernels.hip -------------------------------------------------------------------------
struct ParPlan {
    int n_dil;
    int dil[kMaxDil];
    float posw[kMaxDil * 8];  // softmax of the position prior (models/PAR.py:77,82-85)
};
int par_make_plan(const int *dilations, int n_dil, ParPlan *plan);
// aff [B][NN][h*w] from imgs [B,3,h,w]
int par_launch_affinity(const float *imgs, float *aff, int B, int h, int w, const ParPlan &plan, hipStream_t st);
// one propagation step:  dst[b][k] = sum_n aff[b][n] * gather(src[b][k], n)   for k < kcount[b] (or K if null)
// planes of image b start at b*plane_stride floats.
int par_launch_step(const float *aff, const float *src, float *dst, int B, int Kmax, const int *kcount, int halves,
                    size_t plane_stride, int h, int w, const ParPlan &plan, hipStream_t st);

}  // namespace cosa
