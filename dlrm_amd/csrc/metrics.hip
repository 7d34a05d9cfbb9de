// metrics.hip — device-side evaluation metrics for the inference pass.
//
// Reference call sites replaced (dlrm_s_pytorch.py:759-899, `inference()`):
//   A_test = np.sum((np.round(S_test, 0) == T_test).astype(np.uint8))                      :819-821
//   sklearn.metrics.{recall,precision,f1,accuracy}_score(y_true, np.round(y_score))        :828-847
//   sklearn.metrics.roc_auc_score / average_precision_score(y_true, y_score)               :841-842
// The reference copies every prediction to the host and runs numpy / scikit-learn there; here the
// scores stay in HBM: one pass counts the confusion matrix at the np.round threshold, a radix sort
// (rocPRIM) orders the scores, two scans give cumulative true positives and tie-group starts, and one
// pass over the tie-group ends integrates the ROC curve (trapezoids == sklearn's auc over distinct
// thresholds) and the precision-recall step function (sklearn's average precision), in fp64 with a
// fixed reduction order (deterministic).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/functional.hpp>
#include "common.h"

namespace {

constexpr int kBlk = 256;

struct MetricsLayout {
    size_t keys_in, keys_out, lab_in, lab_out, cum_tp, start, partials, counters, temp, temp_bytes, total;
    int nblk;
};

static size_t m_align(size_t x) { return (x + 255) & ~(size_t)255; }

// descending-score order as an ascending unsigned key: the usual order-preserving float->uint map, inverted.
// -0.0 is folded onto +0.0 (numpy compares them equal: one threshold).
__device__ __forceinline__ unsigned score_key(float s) {
    if (s == 0.f) s = 0.f;
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending in s
    return ~u;                                            // ascending key == descending score
}

// counters: [0] positives  [1] TP  [2] FP  [3] FN  [4] TN  [5] round(score)==target matches
__global__ __launch_bounds__(kBlk) void metrics_prepare_kernel(long long n, const float* __restrict__ s,
                                                               const float* __restrict__ t, unsigned* __restrict__ keys,
                                                               unsigned* __restrict__ lab,
                                                               unsigned long long* __restrict__ counters) {
    __shared__ unsigned red[6][kBlk / 64];
    unsigned c[6] = {0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < n; i += (long long)gridDim.x * kBlk) {
        const float si = s[i], ti = t[i];
        const unsigned y = ti > 0.5f ? 1u : 0u;
        const float r = rintf(si);                         // np.round: half to even
        const unsigned p = r > 0.5f ? 1u : 0u;
        keys[i] = score_key(si);
        lab[i] = y;
        c[0] += y;
        c[1] += (p & y); c[2] += (p & (y ^ 1u)); c[3] += ((p ^ 1u) & y); c[4] += ((p ^ 1u) & (y ^ 1u));
        c[5] += (r == ti) ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long v = 0;
#pragma unroll
        for (int w = 0; w < kBlk / 64; ++w) v += red[threadIdx.x][w];
        if (v) atomicAdd(counters + threadIdx.x, v);       // integer: order independent
    }
}

// group-start index of every sorted position, as the input of a max-scan: i where a new score value begins, else 0
struct HeadIndex {
    const unsigned* keys;
    __device__ unsigned operator()(unsigned i) const { return (i == 0 || keys[i] != keys[i - 1]) ? i : 0u; }
};
struct MaxOp { __device__ unsigned operator()(unsigned a, unsigned b) const { return a > b ? a : b; } };

// one thread per sorted position; only the LAST position of a tie group (one distinct threshold) contributes
__global__ __launch_bounds__(kBlk) void metrics_curve_kernel(long long n, const unsigned* __restrict__ keys,
                                                             const unsigned* __restrict__ cum_tp,
                                                             const unsigned* __restrict__ start,
                                                             double* __restrict__ partials) {
    __shared__ double red[2][kBlk];
    double auc = 0.0, ap = 0.0;
    for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < n; i += (long long)gridDim.x * kBlk) {
        const bool last = (i == n - 1) || keys[i + 1] != keys[i];
        if (!last) continue;
        const unsigned s0 = start[i];
        const double tp = (double)cum_tp[i], tp_prev = s0 ? (double)cum_tp[s0 - 1] : 0.0;
        const double cnt = (double)(i + 1), cnt_prev = (double)s0;
        const double d_tp = tp - tp_prev, d_fp = (cnt - cnt_prev) - d_tp;
        auc += d_fp * (tp + tp_prev) * 0.5;               // trapezoid between consecutive ROC points (unnormalised)
        ap += d_tp * (tp / cnt);                          // (R_k - R_{k-1}) * P_k  (unnormalised by P)
    }
    red[0][threadIdx.x] = auc; red[1][threadIdx.x] = ap;
    __syncthreads();
    for (int k = kBlk / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = red[0][0]; partials[2 * blockIdx.x + 1] = red[1][0]; }
}

__global__ __launch_bounds__(kBlk) void metrics_finish_kernel(long long n, int nblk, const double* __restrict__ partials,
                                                              const unsigned long long* __restrict__ counters,
                                                              double* __restrict__ out) {
    __shared__ double red[2][kBlk];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kBlk) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int k = kBlk / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double P = (double)counters[0], N = (double)n - P;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        out[0] = (double)n; out[1] = P;
        out[2] = (double)counters[1]; out[3] = (double)counters[2]; out[4] = (double)counters[3]; out[5] = (double)counters[4];
        out[6] = (P > 0 && N > 0) ? red[0][0] / (P * N) : nan;     // sklearn raises when only one class is present
        out[7] = (P > 0) ? red[1][0] / P : nan;
        out[8] = (double)counters[5];
    }
}

static int metrics_layout(size_t n, MetricsLayout* lo) {
    size_t o = 0;
    lo->keys_in = o;  o += m_align(n * 4);
    lo->keys_out = o; o += m_align(n * 4);
    lo->lab_in = o;   o += m_align(n * 4);
    lo->lab_out = o;  o += m_align(n * 4);
    lo->cum_tp = o;   o += m_align(n * 4);
    lo->start = o;    o += m_align(n * 4);
    long long nblk = ((long long)n + kBlk - 1) / kBlk; if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
    lo->nblk = (int)nblk;
    lo->partials = o; o += m_align((size_t)nblk * 2 * sizeof(double));
    lo->counters = o; o += m_align(8 * sizeof(unsigned long long));
    size_t t_sort = 0, t_scan1 = 0, t_scan2 = 0;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, const unsigned*, unsigned*, const unsigned*, unsigned*>(
        nullptr, t_sort, nullptr, nullptr, nullptr, nullptr, n, 0, 32, (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    e = rocprim::inclusive_scan(nullptr, t_scan1, (const unsigned*)nullptr, (unsigned*)nullptr, n, rocprim::plus<unsigned>(),
                                (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned>(0u), HeadIndex{nullptr});
    e = rocprim::inclusive_scan(nullptr, t_scan2, heads, (unsigned*)nullptr, n, MaxOp(), (hipStream_t)0, false);
    if (e != hipSuccess) return (int)e;
    size_t t = t_sort; if (t_scan1 > t) t = t_scan1; if (t_scan2 > t) t = t_scan2;
    lo->temp_bytes = t;
    lo->temp = o; o += m_align(t);
    lo->total = o;
    return 0;
}

}  // namespace

extern "C" int64_t dlrm_binary_metrics_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    MetricsLayout lo;
    if (metrics_layout((size_t)n, &lo) != 0) return -1;
    return (int64_t)lo.total;
}

extern "C" int dlrm_binary_metrics(int64_t n, const float* scores, const float* targets, double* out,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (n <= 0 || !scores || !targets || !out || !workspace) return DLRM_E_ARG;
    if (n >= ((int64_t)1 << 32)) return DLRM_E_RANGE;
    if (!dlrm_aligned16(workspace)) return DLRM_E_ALIGN;
    MetricsLayout lo;
    int rc = metrics_layout((size_t)n, &lo);
    if (rc) return rc;
    if ((size_t)workspace_bytes < lo.total) return DLRM_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned* keys_in = (unsigned*)(ws + lo.keys_in), *keys_out = (unsigned*)(ws + lo.keys_out);
    unsigned* lab_in = (unsigned*)(ws + lo.lab_in), *lab_out = (unsigned*)(ws + lo.lab_out);
    unsigned* cum_tp = (unsigned*)(ws + lo.cum_tp), *start = (unsigned*)(ws + lo.start);
    double* partials = (double*)(ws + lo.partials);
    unsigned long long* counters = (unsigned long long*)(ws + lo.counters);
    hipError_t e = hipMemsetAsync(counters, 0, 8 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(metrics_prepare_kernel, dim3(lo.nblk), dim3(kBlk), 0, st, (long long)n, scores, targets, keys_in, lab_in, counters);
    DLRM_LAUNCH_CHECK();
    size_t tb = lo.temp_bytes;
    e = rocprim::radix_sort_pairs(ws + lo.temp, tb, (const unsigned*)keys_in, keys_out, (const unsigned*)lab_in, lab_out,
                                  (size_t)n, 0, 32, st, false);
    if (e != hipSuccess) return (int)e;
    tb = lo.temp_bytes;
    e = rocprim::inclusive_scan(ws + lo.temp, tb, (const unsigned*)lab_out, cum_tp, (size_t)n, rocprim::plus<unsigned>(), st, false);
    if (e != hipSuccess) return (int)e;
    tb = lo.temp_bytes;
    auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned>(0u), HeadIndex{keys_out});
    e = rocprim::inclusive_scan(ws + lo.temp, tb, heads, start, (size_t)n, MaxOp(), st, false);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(metrics_curve_kernel, dim3(lo.nblk), dim3(kBlk), 0, st, (long long)n, (const unsigned*)keys_out,
                       (const unsigned*)cum_tp, (const unsigned*)start, partials);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(1), dim3(kBlk), 0, st, (long long)n, lo.nblk, (const double*)partials,
                       (const unsigned long long*)counters, out);
    DLRM_LAUNCH_CHECK();
    return 0;
}
