// multihot.hip — BASELINE.json configs[4] inputs on the device: the 1-hot -> multi-hot expansion of the MLPerf-v2 synthetic
// data set.
//
// Reference replaced: torchrec_dlrm/multi_hot.py `Multihot`
//   __make_multi_hot_indices_tables (:80-113)  per table t a lookup table M_t [n_t, h_t] int32, column 0 = the id itself,
//                                              columns 1.. = randint(0, n_t) ("uniform") or int32(pareto(0.25)) % n_t
//   __make_new_batch (:129-159)                values_t = F.embedding(ids_t [B], M_t).reshape(-1); values = cat over tables
//   __make_offsets (:115-127)                  offsets = cumsum([0] + [h_t] * B for every table)       (int64, [T*B + 1])
// on the host CPU per batch (and 24 GB of lookup tables for the MLPerf sizes: Σ n_t·h_t·4 B with a 40 M-row, 100-hot table).
//
// Here: the tables live in HBM (288 GB hold them beside the 104 GB of embeddings); one launch expands the [T, B] 1-hot ids
// of a batch into the KJT value layout (table-major, h_t consecutive ids per sample) — consecutive threads write
// consecutive output elements (coalesced 4-byte stores, each lookup row read as one contiguous h_t*4-byte segment) — and
// writes both offset forms: the reference's global cumulative offsets and the per-table local bag starts (b * h_t) that
// dlrm_emb_fwd / dlrm_emb_bwd_* take.  Integer work, HBM-bound: 8 B per produced id.
//
// dlrm_multihot_gen_table builds a lookup table in place from Philox4x32-10 (same distributions as the reference, different
// stream: numpy's MT19937 randint/pareto sequences are not reproduced; tables built by the reference's own class can be
// uploaded instead — the parity tests do that).
#include "common.h"

namespace {

struct Philox {
    unsigned k0, k1;
    __device__ __forceinline__ void block(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) const {
        unsigned c[4] = {c0, c1, c2, c3};
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
            const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ a, n1 = (unsigned)p1;
            const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ b, n3 = (unsigned)p0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {
    return (double)((((unsigned long long)(hi >> 5)) << 26) | (unsigned long long)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// element (r, c) of lookup table `table_id`: c == 0 -> r; else the draw of Philox block (r lo, r hi, c, 0xC0000000 | table_id)
__global__ __launch_bounds__(256) void multihot_gen_kernel(Philox rng, int table_id, long long rows, int hot, int dist,
                                                           int* __restrict__ out) {
    const long long total = rows * hot;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / hot;
        const int c = (int)(e - r * hot);
        if (c == 0) { out[e] = (int)r; continue; }
        unsigned w[4];
        rng.block((unsigned)r, (unsigned)(r >> 32), (unsigned)c, 0xC0000000u | (unsigned)table_id, w);
        if (dist == 0) {
            // uniform on [0, rows): multiply-shift of a 32-bit word (rows < 2^31)
            out[e] = (int)(((unsigned long long)w[0] * (unsigned long long)rows) >> 32);
        } else {
            // numpy pareto(a) = exp(standard_exponential / a) - 1 = (1 - u)^(-1/a) - 1, a = 0.25; .astype(int32) of an
            // out-of-range double gives INT_MIN on x86 (cvttsd2si); numpy's % is a floor-mod (non-negative for rows > 0)
            const double x = exp(-log1p(-u53(w[0], w[1])) * 4.0) - 1.0;
            const long long v = (x < 2147483648.0) ? (long long)x : -2147483648LL;
            long long m = v % rows; if (m < 0) m += rows;
            out[e] = (int)m;
        }
    }
}

#define DLRM_MH_MAX_TABLES 32
struct MhArgs {
    const int* table[DLRM_MH_MAX_TABLES];
    long long  rows[DLRM_MH_MAX_TABLES];
    long long  vbase[DLRM_MH_MAX_TABLES];     // position of table t's first value in `values` (= B * sum_{k<t} hot_k)
    int        hot[DLRM_MH_MAX_TABLES];
    int        slot[DLRM_MH_MAX_TABLES];      // global table number (row of `ids`, block of the offset arrays)
};

template <typename IT>
__global__ __launch_bounds__(256) void multihot_expand_kernel(MhArgs a, long long B, const IT* __restrict__ ids,
                                                              int* __restrict__ values, long long* __restrict__ off_global,
                                                              int* __restrict__ off_local, long long total_values,
                                                              int last_table, long long* err) {
    const int t = blockIdx.y;
    const int h = a.hot[t];
    const int g = a.slot[t];
    const int* __restrict__ tab = a.table[t];
    const long long rows = a.rows[t];
    const long long n = B * h;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long b = e / h;
        const int j = (int)(e - b * h);
        long long id = (long long)ids[(long long)g * B + b];
        if (!dlrm_index_ok(id, rows)) {                      // F.embedding raises on it: keep the id in range, report
            if (j == 0) dlrm_report_bad_index(err, g, id, rows);
            id = 0;
        }
        values[a.vbase[t] + e] = tab[id * h + j];
        if (j == 0) {
            if (off_global) off_global[(long long)g * B + b] = a.vbase[t] + e;
            if (off_local) off_local[(long long)g * B + b] = (int)e;
        }
    }
    if (off_global && t == last_table && blockIdx.x == 0 && threadIdx.x == 0) off_global[((long long)g + 1) * B] = total_values;
}

}  // namespace

extern "C" int dlrm_multihot_gen_table(int table_id, int64_t rows, int hot, int dist, uint64_t seed, int32_t* table_out,
                                       void* stream) {
    if (table_id < 0 || rows <= 0 || rows >= (1LL << 31) || hot <= 0 || !table_out) return DLRM_E_ARG;
    if (dist != 0 && dist != 1) return DLRM_E_MODE;
    Philox rng = {(unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32)};
    long long nblk = (rows * hot + 255) / 256; if (nblk > 65536) nblk = 65536;
    hipLaunchKernelGGL(multihot_gen_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, rng, table_id,
                       (long long)rows, hot, dist, table_out);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_multihot_expand(int T, int64_t B, const void* ids, int idx_bits, const void* const* tables_host,
                                    const int64_t* rows_host, const int* hot_host, int32_t* values, int64_t* offsets_global,
                                    int32_t* offsets_local, int64_t* err, void* stream) {
    if (T <= 0 || B <= 0 || !ids || !tables_host || !rows_host || !hot_host || !values) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    long long total = 0;
    for (int t = 0; t < T; ++t) {
        if (!tables_host[t] || rows_host[t] <= 0 || hot_host[t] <= 0) return DLRM_E_ARG;
        total += (long long)B * hot_host[t];
    }
    if (total >= (1LL << 31)) return DLRM_E_RANGE;           // values / local offsets are int32 positions, like the reference's int32 KJT
    long long base = 0;
    for (int t0 = 0; t0 < T; t0 += DLRM_MH_MAX_TABLES) {
        const int n = (T - t0 < DLRM_MH_MAX_TABLES) ? T - t0 : DLRM_MH_MAX_TABLES;
        MhArgs a = {};
        int max_hot = 1;
        for (int k = 0; k < n; ++k) {
            const int t = t0 + k;
            a.table[k] = (const int*)tables_host[t]; a.rows[k] = rows_host[t]; a.hot[k] = hot_host[t]; a.slot[k] = t;
            a.vbase[k] = base; base += (long long)B * hot_host[t];
            if (hot_host[t] > max_hot) max_hot = hot_host[t];
        }
        long long nblk = (B * max_hot + 255) / 256; if (nblk > 4096) nblk = 4096;
        dim3 grid((unsigned)nblk, (unsigned)n, 1), block(256);
        const int last = (t0 + n == T) ? n - 1 : -1;
        if (idx_bits == 64)
            hipLaunchKernelGGL(multihot_expand_kernel<long long>, grid, block, 0, (hipStream_t)stream, a, (long long)B,
                               (const long long*)ids, values, (long long*)offsets_global, offsets_local, total, last, (long long*)err);
        else
            hipLaunchKernelGGL(multihot_expand_kernel<int>, grid, block, 0, (hipStream_t)stream, a, (long long)B,
                               (const int*)ids, values, (long long*)offsets_global, offsets_local, total, last, (long long*)err);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}
