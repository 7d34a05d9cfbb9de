// datagen.hip — the reference's synthetic ("random") input batches, generated in HBM.
//
// Reference replaced: generate_dist_input_batch with rand_data_dist == "uniform" + generate_random_output_batch
// (dlrm_data_pytorch.py:899-960, 835-846), which build one Python list per bag on the host (0.4 s per 2048-sample
// batch, ~13 s at the MLPerf batch of 65536):
//   X[b, :]        = rand(m_den) as float32
//   for table t, bag b:  L = P if fixed else round(max(1, r * min(rows_t, P)))        r ~ U[0,1)
//                        indices = unique(round(r_j * (rows_t - 1)))  j < L            (sorted, duplicates removed)
//   offsets[t][b]  = running sum of the (post-unique) bag lengths
//   target[b]      = round(rand()) (or rand() when targets are not rounded)
// Same distributions, different stream: numpy's MT19937 sequence is replaced by Philox4x32-10 keyed by
// (seed, table, bag) so that every bag is an independent counter-based draw (order-free, reproducible).
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include "common.h"

namespace {

constexpr int kMaxLookups = 128;         // compiled limit of num_indices_per_lookup

struct Philox {
    unsigned k0, k1;
    __device__ __forceinline__ static void round_(unsigned (&c)[4], unsigned a, unsigned b) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ a, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ b, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ void block(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) const {
        unsigned c[4] = {c0, c1, c2, c3};
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) { round_(c, a, b); a += 0x9E3779B9u; b += 0xBB67AE85u; }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// 53-bit uniform double in [0,1) from two 32-bit words (what numpy's random() produces from its generator)
__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {
    return (double)((((unsigned long long)(hi >> 5)) << 26) | (unsigned long long)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

struct GenArgs {
    long long rows[DLRM_MAX_TABLES_PER_LAUNCH];
    void*     off[DLRM_MAX_TABLES_PER_LAUNCH];
    void*     idx[DLRM_MAX_TABLES_PER_LAUNCH];
};

// the sorted, de-duplicated index list of bag (t, b); returns its length
__device__ int make_bag(const Philox& rng, int t, long long b, long long rows, int P, int fixed, long long (&v)[kMaxLookups]) {
    unsigned w[4];
    int L = P;
    if (!fixed) {
        rng.block((unsigned)b, (unsigned)(b >> 32), (unsigned)t, 0x80000000u, w);
        const double lim = (double)(rows < (long long)P ? rows : (long long)P);
        double x = u53(w[0], w[1]) * lim;
        if (x < 1.0) x = 1.0;
        L = (int)rint(x);
    }
    int n = 0;
    for (int j = 0; j < L; j += 2) {
        rng.block((unsigned)b, (unsigned)(b >> 32), (unsigned)t, (unsigned)(j >> 1), w);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (j + h >= L) break;
            const long long val = (long long)rint(u53(w[2 * h], w[2 * h + 1]) * (double)(rows - 1));
            // insertion into the sorted unique prefix v[0..n)
            int p = n;
            while (p > 0 && v[p - 1] > val) --p;
            if (p > 0 && v[p - 1] == val) continue;
            for (int q = n; q > p; --q) v[q] = v[q - 1];
            v[p] = val;
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(256) void gen_lengths_kernel(GenArgs a, Philox rng, long long B, int P, int fixed,
                                                          unsigned* __restrict__ len /* [T][B] */) {
    const int t = blockIdx.y;
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    long long v[kMaxLookups];
    len[(long long)t * B + b] = (unsigned)make_bag(rng, t, b, a.rows[t], P, fixed, v);
}

template <typename IT>
__global__ __launch_bounds__(256) void gen_fill_kernel(GenArgs a, Philox rng, long long B, int P, int fixed,
                                                       const unsigned* __restrict__ scan /* exclusive, [T][B] flattened */,
                                                       long long* __restrict__ nnz_out /* [T] */, unsigned total_last_len_dummy) {
    (void)total_last_len_dummy;
    const int t = blockIdx.y;
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    long long v[kMaxLookups];
    const int n = make_bag(rng, t, b, a.rows[t], P, fixed, v);
    const long long base = (long long)scan[(long long)t * B];
    const long long o = (long long)scan[(long long)t * B + b] - base;
    IT* __restrict__ off = (IT*)a.off[t];
    IT* __restrict__ idx = (IT*)a.idx[t];
    off[b] = (IT)o;
    for (int j = 0; j < n; ++j) idx[o + j] = (IT)v[j];
    if (b == B - 1) nnz_out[t] = o + n;
}

// one-hot fast path (P == 1, fixed): offsets = arange, one index per bag, no scan
template <typename IT>
__global__ __launch_bounds__(256) void gen_onehot_kernel(GenArgs a, Philox rng, long long B, long long* __restrict__ nnz_out) {
    const int t = blockIdx.y;
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    unsigned w[4];
    rng.block((unsigned)b, (unsigned)(b >> 32), (unsigned)t, 0u, w);
    ((IT*)a.off[t])[b] = (IT)b;
    ((IT*)a.idx[t])[b] = (IT)(long long)rint(u53(w[0], w[1]) * (double)(a.rows[t] - 1));
    if (b == B - 1) nnz_out[t] = B;
}

__global__ __launch_bounds__(256) void gen_dense_kernel(Philox rng, long long n, float* __restrict__ x, int round_it) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // two values per thread
    if (2 * i >= n) return;
    unsigned w[4];
    rng.block((unsigned)i, (unsigned)(i >> 32), 0x7fffffffu, 0x40000000u, w);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (2 * i + h >= n) break;
        const float f = (float)u53(w[2 * h], w[2 * h + 1]);                   // rand().astype(np.float32)
        x[2 * i + h] = round_it ? rintf(f) : f;
    }
}

}  // namespace

extern "C" int64_t dlrm_gen_workspace_bytes(int T, int64_t B) {
    if (T <= 0 || B <= 0) return 0;
    const size_t n = (size_t)T * (size_t)B;
    size_t tb = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tb, (const unsigned*)nullptr, (unsigned*)nullptr, 0u, n,
                                           rocprim::plus<unsigned>(), (hipStream_t)0, false);
    if (e != hipSuccess) return -1;
    const size_t a = (n * 4 + 255) & ~(size_t)255;
    return (int64_t)(2 * a + ((tb + 255) & ~(size_t)255));
}

extern "C" int dlrm_gen_uniform_bags(int T, int64_t B, const int64_t* rows_host, int num_indices_per_lookup, int fixed,
                                     uint64_t seed, int idx_bits, void* const* offsets_host, void* const* indices_host,
                                     int64_t* nnz_dev, void* workspace, int64_t workspace_bytes, void* stream) {
    if (T <= 0 || T > DLRM_MAX_TABLES_PER_LAUNCH || B <= 0 || !rows_host || !offsets_host || !indices_host || !nnz_dev)
        return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    const int P = num_indices_per_lookup;
    if (P < 1 || P > kMaxLookups) return DLRM_E_RANGE;
    if ((int64_t)T * B * P >= ((int64_t)1 << 32)) return DLRM_E_RANGE;       // 32-bit scan of the lengths
    hipStream_t st = (hipStream_t)stream;
    GenArgs a = {};
    for (int t = 0; t < T; ++t) {
        if (rows_host[t] < 1 || !offsets_host[t] || !indices_host[t]) return DLRM_E_ARG;
        a.rows[t] = rows_host[t]; a.off[t] = offsets_host[t]; a.idx[t] = indices_host[t];
    }
    const Philox rng = {(unsigned)seed, (unsigned)(seed >> 32)};
    dim3 grid((unsigned)((B + 255) / 256), (unsigned)T, 1), block(256);
    if (P == 1 && fixed) {
        if (idx_bits == 64) hipLaunchKernelGGL(gen_onehot_kernel<long long>, grid, block, 0, st, a, rng, (long long)B, (long long*)nnz_dev);
        else                hipLaunchKernelGGL(gen_onehot_kernel<int>, grid, block, 0, st, a, rng, (long long)B, (long long*)nnz_dev);
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    const int64_t need = dlrm_gen_workspace_bytes(T, B);
    if (need < 0) return DLRM_E_ARG;
    if (!workspace || !dlrm_aligned16(workspace) || workspace_bytes < need) return DLRM_E_ARG;
    const size_t n = (size_t)T * (size_t)B;
    const size_t abytes = (n * 4 + 255) & ~(size_t)255;
    unsigned* len = (unsigned*)workspace;
    unsigned* scan = (unsigned*)((char*)workspace + abytes);
    void* temp = (char*)workspace + 2 * abytes;
    size_t tb = (size_t)workspace_bytes - 2 * abytes;
    hipLaunchKernelGGL(gen_lengths_kernel, grid, block, 0, st, a, rng, (long long)B, P, fixed ? 1 : 0, len);
    DLRM_LAUNCH_CHECK();
    hipError_t e = rocprim::exclusive_scan(temp, tb, (const unsigned*)len, scan, 0u, n, rocprim::plus<unsigned>(), st, false);
    if (e != hipSuccess) return (int)e;
    if (idx_bits == 64)
        hipLaunchKernelGGL(gen_fill_kernel<long long>, grid, block, 0, st, a, rng, (long long)B, P, fixed ? 1 : 0,
                           (const unsigned*)scan, (long long*)nnz_dev, 0u);
    else
        hipLaunchKernelGGL(gen_fill_kernel<int>, grid, block, 0, st, a, rng, (long long)B, P, fixed ? 1 : 0,
                           (const unsigned*)scan, (long long*)nnz_dev, 0u);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_gen_uniform_dense(int64_t n, float* x, int round_values, uint64_t seed, void* stream) {
    if (n <= 0 || !x) return DLRM_E_ARG;
    const Philox rng = {(unsigned)seed, (unsigned)(seed >> 32)};
    const long long threads = (n + 1) / 2;
    hipLaunchKernelGGL(gen_dense_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rng,
                       (long long)n, x, round_values ? 1 : 0);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Criteo-Terabyte binary records -> model inputs, on the device.
// Reference replaced: CriteoBinDataset.__getitem__ + _transform_features (data_loader_terabyte.py:233-248, 74-93): one
// record = 40 little-endian int32 = [label | 13 dense counts | 26 categorical ids]; the reference converts on the host
// (torch.log(x_int + 1), x_cat % max_ind_range, transposes to [26, B], builds offsets arange per table).  Here the raw
// [B, 40] block is copied to HBM once (42 % of the bytes of the converted batch) and one kernel writes every input.
//   X[b, j]      = logf(float(raw[b, 1 + j]) + 1)                 j < 13      (fp32, like torch.log on a float tensor)
//   idx[t, b]    = raw[b, 14 + t] mod max_ind_range (if > 0)       t < 26      (python/torch `%`: result in [0, m))
//   off[t, b]    = b
//   target[b]    = float(raw[b, 0])
// ------------------------------------------------------------------------------------------------------------------
namespace {

template <typename IT>
__global__ __launch_bounds__(256) void criteo_bin_transform_kernel(long long B, const int* __restrict__ raw, long long max_ind_range,
                                                                   float* __restrict__ X, long long ldx, IT* __restrict__ idx,
                                                                   IT* __restrict__ off, long long ld_idx,
                                                                   float* __restrict__ target) {
    // a workgroup owns 64 records: the 64 x 40 block goes through LDS so that both the row-major reads and the
    // table-major (transposed) writes are coalesced
    __shared__ int tile[64][41];
    const long long b0 = (long long)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 40; e += 256) {
        const int r = e / 40, c = e - r * 40;
        tile[r][c] = (b0 + r < B) ? raw[(b0 + r) * 40 + c] : 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 13; e += 256) {              // dense features, row-major output
        const int r = e / 13, j = e - r * 13;
        if (b0 + r < B) X[(b0 + r) * ldx + j] = logf((float)tile[r][1 + j] + 1.0f);
    }
    for (int e = threadIdx.x; e < 26 * 64; e += 256) {              // categorical ids, table-major output
        const int t = e >> 6, r = e & 63;
        if (b0 + r < B) {
            long long v = (long long)tile[r][14 + t];
            if (max_ind_range > 0) { v %= max_ind_range; if (v < 0) v += max_ind_range; }
            idx[(long long)t * ld_idx + b0 + r] = (IT)v;
            off[(long long)t * ld_idx + b0 + r] = (IT)(b0 + r);
        }
    }
    if (threadIdx.x < 64 && b0 + threadIdx.x < B) target[b0 + threadIdx.x] = (float)tile[threadIdx.x][0];
}

}  // namespace

extern "C" int dlrm_criteo_bin_transform(int64_t B, const int32_t* raw, int64_t max_ind_range, int idx_bits, float* X,
                                         int64_t ldx, void* indices, void* offsets, int64_t ld_idx, float* target,
                                         void* stream) {
    if (B <= 0 || !raw || !X || !indices || !offsets || !target || ldx < 13 || ld_idx < B) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    dim3 grid((unsigned)((B + 63) / 64)), block(256);
    if (idx_bits == 64)
        hipLaunchKernelGGL(criteo_bin_transform_kernel<long long>, grid, block, 0, (hipStream_t)stream, (long long)B, (const int*)raw,
                           (long long)max_ind_range, X, (long long)ldx, (long long*)indices, (long long*)offsets, (long long)ld_idx, target);
    else
        hipLaunchKernelGGL(criteo_bin_transform_kernel<int>, grid, block, 0, (hipStream_t)stream, (long long)B, (const int*)raw,
                           (long long)max_ind_range, X, (long long)ldx, (int*)indices, (int*)offsets, (long long)ld_idx, target);
    DLRM_LAUNCH_CHECK();
    return 0;
}
