// gmm_kernels.hip -- adaptive thresholds (SURVEY f-4): the 1-D Gaussian-mixture fit of the training loop, on the device.
//
//   main.py:138-151,174-184          queue of per-cell CAM maxima -> rungmm -> EMA of (low, high) thresholds, every iteration
//   utils/seg_helper.py:924-943      rungmm: GaussianMixture(2|3, weights/means/precisions given).fit_predict on queue > filter_thre
//
// The reference copies the CAM maxima to the host and runs scikit-learn's EM on ~10^6 samples every iteration (seconds, and
// a host sync).  Here the whole fit -- EM iterations, convergence test, final assignment, max / min of the outer components --
// is ONE launch that never touches the host.  scikit-learn's procedure for one feature (every matrix is 1x1), float64:
//
//   E:  lp_k = -0.5 (log 2pi + ((x - mu_k) pc_k)^2) + log pc_k + log w_k;  lpn = logsumexp_k lp_k;  r_k = exp(lp_k - lpn)
//   M:  n_k = sum r_k + 10 eps;  mu_k = sum r_k x / n_k;  var_k = sum r_k (x - mu_k)^2 / n_k + 1e-6;  pc_k = 1/sqrt(var_k)
//       w_k = n_k / sum_j n_j
//   stop when |mean(lpn) - previous| < tol (that iteration's M step is kept), at most max_iter iterations; labels = argmax_k
//   of one more E step.
//
// Parallel form: kGmmBlocks workgroups stride over the samples; one sweep per iteration gathers sum r, sum r x, sum r x^2 and
// sum lpn (scikit-learn sums r (x - mu_new)^2 in a second sweep; Sxx - 2 mu Sx + mu^2 S0 is the same quantity and saves a
// sweep and a barrier), ending in per-workgroup partials, a grid barrier, and every workgroup adding the partials up in the
// same fixed order -- so all workgroups hold bit-identical parameters, take
// the same branch at the convergence test, and the fit is deterministic run to run.  The grid barrier is a monotonic counter;
// its spin is bounded (status bit 8 on expiry) so every wave reaches the end of the kernel whatever happens.
#include "kernels.hpp"

#include <cfloat>
#include <cstdlib>

namespace cosa {
namespace {

constexpr int kGmmBlocks = 128;             // measured: 64 -> 271 us, 128 -> 205, 256 -> 241, 512 -> 472 per fit (the barrier grows with the grid)
constexpr int kGmmThreads = 256;
constexpr int kGmmCols = 16;                // doubles per workgroup partial row (3K + 1 <= 10 used)
constexpr double kLog2Pi = 1.8378770664093453;

struct GmmShared {
    double wave[kGmmThreads / 64][kGmmCols];
    double group[16][kGmmCols];
    double total[kGmmCols];
};

// bounded grid barrier: `target` = arrivals expected so far (monotonic counter, zeroed by the host before the launch)
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target)
{
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        int good = 0;
        for (int spin = 0; spin < (1 << 22); spin++) {
            if (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// sum `v[0..ncol)` over the workgroup, write the row of this workgroup, barrier, then add the rows of all workgroups in
// index order (thread c owns column c); the totals land in sh.total for every thread of every workgroup.
template <int NCOL>
__device__ __forceinline__ bool all_reduce(double (&v)[NCOL], GmmShared &sh, double *__restrict__ rows, unsigned *counter,
                                           unsigned &arrivals)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NCOL; c++) {
        double s = v[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) sh.wave[wv][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < NCOL) {
        double s = sh.wave[0][threadIdx.x];
        for (int k = 1; k < kGmmThreads / 64; k++) s += sh.wave[k][threadIdx.x];
        rows[(size_t)blockIdx.x * kGmmCols + threadIdx.x] = s;
        __threadfence();                                             // the row is at the L2 before this workgroup arrives
    }
    arrivals += gridDim.x;
    const bool ok = grid_barrier(counter, arrivals);
    // totals: thread t adds rows (t / 16), (t / 16) + 16, ... of column t % 16 (independent loads, few per thread), then thread c
    // adds the 16 group sums of column c in index order.  The same fixed tree in every workgroup: identical totals everywhere.
    {
        static_assert(kGmmCols == 16 && kGmmThreads == 256, "reduction layout");
        const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
        double s = 0.0;
        if (c < NCOL)
            for (unsigned g = grp; g < gridDim.x; g += 16)
                s += __hip_atomic_load(rows + (size_t)g * kGmmCols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.group[grp][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < NCOL) {
        double s = sh.group[0][threadIdx.x];
        for (int g = 1; g < 16; g++) s += sh.group[g][threadIdx.x];
        sh.total[threadIdx.x] = s;
    }
    __syncthreads();
    return ok;
}

template <int K>
__device__ __forceinline__ double e_step(double x, const double (&mu)[K], const double (&pc)[K], const double (&lpc)[K],
                                         const double (&lw)[K], double (&lp)[K])
{
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const double y = x * pc[k] - mu[k] * pc[k];
        lp[k] = (-0.5 * (kLog2Pi + y * y) + lpc[k]) + lw[k];         // lpc = log pc, lw = log w (scikit-learn's order of additions)
        m = lp[k] > m ? lp[k] : m;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) s += exp(lp[k] - m);
    return log(s) + m;
}

// out: [0] max of component 0, [1] min of component K-1 (K = 3), [2] iterations, [3] status bits (0 = fine; 1 component 0 empty;
// 2 component 2 empty; 4 fewer samples than components; 8 barrier expired), [4..4+K) means, [7..7+K) weights, [10..10+K) pc
template <int K>
__global__ __launch_bounds__(kGmmThreads) void gmm_fit_kernel(const double *__restrict__ xs, const long long *__restrict__ n_ptr,
                                                             double tol, double reg, int max_iter, double *__restrict__ rows,
                                                             unsigned *__restrict__ counter, unsigned long long *__restrict__ extrema,
                                                             double *__restrict__ out)
{
    __shared__ GmmShared sh;
    const long long n = *n_ptr;
    if (n < K) {                                                     // uniform over the grid
        if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = out[1] = NAN; out[2] = 0.0; out[3] = 4.0; }
        return;
    }
    double mu[K], pc[K], w[K], lw[K], lpc[K];
    mu[0] = xs[0];
    mu[K - 1] = xs[n - 1];
    if (K == 3) mu[1] = (xs[(n - 1) / 2] + xs[n / 2]) / 2.0;        // np.median of the sorted samples
#pragma unroll
    for (int k = 0; k < K; k++) { w[k] = 1.0 / (double)K; pc[k] = 1.0; }
    double lb = -INFINITY;
    unsigned arrivals = 0;
    bool ok = true;
    int it = 0;
    const long long stride = (long long)gridDim.x * kGmmThreads;
    const long long first = (long long)blockIdx.x * kGmmThreads + threadIdx.x;
    for (it = 1; it <= max_iter; it++) {
#pragma unroll
        for (int k = 0; k < K; k++) { lpc[k] = log(pc[k]); lw[k] = log(w[k]); }
        // one sweep: sum r_k, sum r_k x, sum r_k x^2, sum lpn
        double a[3 * K + 1];
#pragma unroll
        for (int c = 0; c < 3 * K + 1; c++) a[c] = 0.0;
        for (long long i = first; i < n; i += stride) {
            const double x = xs[i];
            double lp[K];
            const double lpn = e_step<K>(x, mu, pc, lpc, lw, lp);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const double r = exp(lp[k] - lpn);
                const double rx = r * x;
                a[k] += r;
                a[K + k] += rx;
                a[2 * K + k] += rx * x;
            }
            a[3 * K] += lpn;
        }
        // partial rows alternate between two buffers: a workgroup that is already in iteration it+1 must not overwrite rows that a
        // slower one is still adding up for iteration it
        ok = all_reduce<3 * K + 1>(a, sh, rows + (size_t)(it & 1) * kGmmBlocks * kGmmCols, counter, arrivals) && ok;
        double nsum = 0.0, nk[K];
#pragma unroll
        for (int k = 0; k < K; k++) { nk[k] = sh.total[k] + 10.0 * DBL_EPSILON; nsum += nk[k]; }
#pragma unroll
        for (int k = 0; k < K; k++) {
            // sum r (x - m)^2 = Sxx - 2 m Sx + m^2 S0 with the new mean m: the quantity scikit-learn sums in a second sweep
            // (equal up to ~1e-14 relative; the thresholds are samples, not these parameters)
            const double m = sh.total[K + k] / nk[k];
            const double ssd = sh.total[2 * K + k] - 2.0 * m * sh.total[K + k] + m * m * sh.total[k];
            const double var = (ssd > 0.0 ? ssd : 0.0) / nk[k] + reg;
            mu[k] = m;
            w[k] = nk[k] / nsum;
            pc[k] = 1.0 / sqrt(var);
        }
        const double lbn = sh.total[3 * K] / (double)n;
        __syncthreads();                                             // sh.total is rewritten by the next reduction
        const double change = lbn - lb;
        lb = lbn;
        if (fabs(change) < tol || !ok) break;
    }
    if (it > max_iter) it = max_iter;
    // assignment with the final parameters; extrema of the outer components (positive doubles order like their bit patterns)
#pragma unroll
    for (int k = 0; k < K; k++) { lpc[k] = log(pc[k]); lw[k] = log(w[k]); }
    // the minimum is kept as the maximum of the complemented bits, so that "nothing yet" is 0 for both and one memset arms them
    unsigned long long lo_max = 0ull, hi_min_c = 0ull;
    for (long long i = first; i < n; i += stride) {
        const double x = xs[i];
        double lp[K];
        const double lpn = e_step<K>(x, mu, pc, lpc, lw, lp);
        int best = 0;
        double bv = lp[0] - lpn;
#pragma unroll
        for (int k = 1; k < K; k++) {
            const double q = lp[k] - lpn;
            if (q > bv) { bv = q; best = k; }
        }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
        if (best == 0) lo_max = bits > lo_max ? bits : lo_max;
        if (best == 2) hi_min_c = ~bits > hi_min_c ? ~bits : hi_min_c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_down(lo_max, o, 64), b = __shfl_down(hi_min_c, o, 64);
        lo_max = a > lo_max ? a : lo_max;
        hi_min_c = b > hi_min_c ? b : hi_min_c;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(extrema, lo_max);
        atomicMax(extrema + 1, hi_min_c);
    }
    arrivals += gridDim.x;
    ok = grid_barrier(counter, arrivals) && ok;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long a = __hip_atomic_load(extrema, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(extrema + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int status = ok ? 0 : 8;
        if (a == 0ull) status |= 1;
        if (K == 3 && b == 0ull) status |= 2;
        out[0] = a == 0ull ? NAN : __longlong_as_double((long long)a);
        out[1] = (K == 3 && b != 0ull) ? __longlong_as_double((long long)~b) : NAN;
        out[2] = (double)it;
        out[3] = (double)status;
        for (int k = 0; k < K; k++) { out[4 + k] = mu[k]; out[7 + k] = w[k]; out[10 + k] = pc[k]; }
    }
}

}  // namespace
}  // namespace cosa

using namespace cosa;

extern "C" size_t cosa_gmm_workspace_bytes(void)
{
    return align_up((size_t)2 * kGmmBlocks * kGmmCols * sizeof(double), 256) + 256;
}

// sorted: the samples above the filter threshold in ascending order (device, float64, positive); n_dev: their count (device int64,
// <= capacity); out: 13 device doubles (see the kernel).  No host synchronisation.
extern "C" int cosa_gmm_fit_thresholds(const double *sorted, const long long *n_dev, long long capacity, int modal, double tol,
                                       double reg_covar, int max_iter, double *out, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    COSA_REQUIRE(sorted && n_dev && out && workspace, "cosa_gmm_fit_thresholds: null pointer");
    COSA_REQUIRE(modal == 2 || modal == 3, "cosa_gmm_fit_thresholds: modal must be 2 or 3");
    COSA_REQUIRE(capacity > 0 && max_iter > 0 && tol > 0.0 && reg_covar >= 0.0, "cosa_gmm_fit_thresholds: bad arguments");
    if (workspace_bytes < cosa_gmm_workspace_bytes()) {
        set_error("cosa_gmm_fit_thresholds: workspace too small");
        return COSA_ENOMEM;
    }
    hipStream_t st = as_stream(stream);
    Carver cv(workspace);
    double *rows = cv.take<double>((size_t)2 * kGmmBlocks * kGmmCols);
    unsigned long long *tail = cv.take<unsigned long long>(4);       // [0..1] extrema, [2] barrier counter
    COSA_HIP_CHECK(hipMemsetAsync(tail, 0, 4 * sizeof(unsigned long long), st));
    // a small grid when there are few samples: every workgroup must be resident for the barrier (128 x 256 threads, 2.4 KB of LDS: always are)
    long long want = (capacity + kGmmThreads - 1) / kGmmThreads;
    const int blocks = (int)(want < 1 ? 1 : (want > kGmmBlocks ? kGmmBlocks : want));
    unsigned *counter = reinterpret_cast<unsigned *>(tail + 2);
    if (modal == 3)
        hipLaunchKernelGGL(gmm_fit_kernel<3>, dim3(blocks), dim3(kGmmThreads), 0, st, sorted, n_dev, tol, reg_covar, max_iter, rows,
                           counter, tail, out);
    else
        hipLaunchKernelGGL(gmm_fit_kernel<2>, dim3(blocks), dim3(kGmmThreads), 0, st, sorted, n_dev, tol, reg_covar, max_iter, rows,
                           counter, tail, out);
    COSA_LAUNCH_CHECK();
    return COSA_OK;
}
