// sorted_common.h — pieces shared by the sort-based embedding updates (emb_sorted.hip: SGD, adagrad.hip: row-wise
// Adagrad): vector helpers, the (table,row) key expansion kernel, and the workspace layout around rocPRIM's radix sort.
#pragma once
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "seg_sort.h"

namespace {


template <int VEC> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };

__device__ __forceinline__ void v_zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void v_zero(float& a) { a = 0.f; }
__device__ __forceinline__ void v_fma(float4& a, float w, const float4& v) {
    a.x = __builtin_fmaf(w, v.x, a.x); a.y = __builtin_fmaf(w, v.y, a.y);
    a.z = __builtin_fmaf(w, v.z, a.z); a.w = __builtin_fmaf(w, v.w, a.w);
}
__device__ __forceinline__ void v_fma(float& a, float w, const float& v) { a = __builtin_fmaf(w, v, a); }
__device__ __forceinline__ float4 v_mul(float s, const float4& v) { return make_float4(s * v.x, s * v.y, s * v.z, s * v.w); }
__device__ __forceinline__ float v_mul(float s, const float& v) { return s * v; }
__device__ __forceinline__ void v_atomic_add(float* p, const float4& v) {
    atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
__device__ __forceinline__ void v_atomic_add(float* p, const float& v) { atomicAdd(p, v); }

// Table rows through GLOBAL pointers.  A table's address reaches the update kernels as an integer in an LDS table (SortedArgs by value ->
// shared array), and a pointer rebuilt from an integer is GENERIC: its loads / stores / atomics compile to FLAT instructions, which count on
// lgkmcnt as well as vmcnt — the next chunk's first LDS read (s_w[t] ...) then waits for the previous chunk's row stores and atomics to be
// acknowledged by memory.  With an explicit address space they are global_load / global_store / global_atomic_add_f32 (vmcnt only).
typedef float sc_floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float sc_gfloat;
typedef __attribute__((address_space(1))) sc_floatx4 sc_gfloatx4;
__device__ __forceinline__ void v_gload(float4& d, const float* p) { const sc_floatx4 v = *(const sc_gfloatx4*)p; d = make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void v_gload(float& d, const float* p) { d = *(const sc_gfloat*)p; }
__device__ __forceinline__ void v_gstore(float* p, const float4& v) { *(sc_gfloatx4*)p = (sc_floatx4){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void v_gstore(float* p, const float& v) { *(sc_gfloat*)p = v; }
__device__ __forceinline__ void g_atomic_add1(float* p, float v) {
    (void)__hip_atomic_fetch_add((sc_gfloat*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void v_gatomic_add(float* p, const float4& v) {
    g_atomic_add1(p + 0, v.x); g_atomic_add1(p + 1, v.y); g_atomic_add1(p + 2, v.z); g_atomic_add1(p + 3, v.w);
}
__device__ __forceinline__ void v_gatomic_add(float* p, const float& v) { g_atomic_add1(p, v); }

struct SortedArgs {
    float*       w[DLRM_MAX_TABLES_PER_LAUNCH];
    const float* psw[DLRM_MAX_TABLES_PER_LAUNCH];
    long long    base[DLRM_MAX_TABLES_PER_LAUNCH];   // first global lookup position of the table
    int          slot[DLRM_MAX_TABLES_PER_LAUNCH];   // dout column block of the table
};

// (table, bag, lookup) -> key = table << row_bits | row, val = global lookup position, bag_of[pos] = bag
template <typename IT, typename KT>
__global__ __launch_bounds__(256) void expand_kernel(EmbArgs a, SortedArgs sa, long long B, int row_bits,
                                                     KT* __restrict__ keys, unsigned* __restrict__ vals,
                                                     unsigned* __restrict__ bag_of, unsigned* __restrict__ zero_words = nullptr) {
    const int t = blockIdx.y;
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const long long nnz = a.nnz[t];
    const long long base = sa.base[t];
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (zero_words && t == 0) zero_words[b] = 0u;        // (dlrm_emb_presort: the per-bag mask single_mask_kernel ORs into after the sort)
    const long long s = (long long)off[b];
    const long long e = (b + 1 < B) ? (long long)off[b + 1] : nnz;
    for (long long i = s; i < e; ++i) {
        const long long pos = base + i;
        long long r = (long long)idx[i];
        unsigned bag = (unsigned)b;
        if (!dlrm_index_ok(r, a.rows[t])) {
            // skipped lookup: it sorts into row 0 of its table with a zero gradient (DLRM_DEAD_BAG) and is reported —
            // an index >= 2^row_bits would otherwise spill into the table bits of the key
            dlrm_report_bad_index(a.err, a.slot[t], r, a.rows[t]);
            r = 0; bag = DLRM_DEAD_BAG;
        }
        keys[pos] = ((KT)t << row_bits) | (KT)r;
        if (vals) vals[pos] = (unsigned)pos;      // (the segmented sorter's first round takes the position itself: vals == nullptr)
        bag_of[pos] = bag;
    }
}

// The same expansion for multi-hot batches, one thread per LOOKUP: with one thread per bag, neighbouring lanes write `hot` entries apart
// (0.54 ms for the 14 M lookups of the MLPerf-v2 batch, 0.4 TB/s); here every store is coalesced and the bag of a position is found by a
// binary search over the table's offsets (256 KB per table at B = 65536: cache-resident, and neighbouring positions walk the same path).
template <typename IT, typename KT>
__global__ __launch_bounds__(256) void expand_positions_kernel(EmbArgs a, SortedArgs sa, long long B, int row_bits,
                                                               KT* __restrict__ keys, unsigned* __restrict__ vals,
                                                               unsigned* __restrict__ bag_of) {
    const int t = blockIdx.y;
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const long long nnz = a.nnz[t];
    const long long base = sa.base[t];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (long long)gridDim.x * 256) {
        // owner of position i = the LAST bag whose start is <= i (empty bags share their start with the bag that follows them and own
        // nothing; off[0] == 0 as EmbeddingBag requires).  Multi-hot batches of the reference have a FIXED number of lookups per bag
        // (torchrec_dlrm/multi_hot.py:80-159), so the proportional guess i * B / nnz is the owner: two loads confirm it instead of the
        // 17 dependent loads of the binary search (228 -> ~90 us for the 14 M lookups of the MLPerf-v2 batch); ragged bags fall through
        long long lo = (long long)(((unsigned long long)i * (unsigned long long)B) / (unsigned long long)(nnz > 0 ? nnz : 1)), hi = B - 1;   // (i < 2^32, B < 2^32)
        if (lo > hi) lo = hi;
        if (!((long long)off[lo] <= i && (lo + 1 >= B || (long long)off[lo + 1] > i))) {
            lo = 0;
            while (lo < hi) {
                const long long mid = (lo + hi + 1) >> 1;
                if ((long long)off[mid] <= i) lo = mid; else hi = mid - 1;
            }
        }
        long long r = (long long)idx[i];
        unsigned bag = (unsigned)lo;
        if (i < (long long)off[0]) { r = 0; bag = DLRM_DEAD_BAG; }           // in front of the first bag: belongs to no bag (EmbeddingBag requires off[0] == 0)
        else if (!dlrm_index_ok(r, a.rows[t])) {
            dlrm_report_bad_index(a.err, a.slot[t], r, a.rows[t]);
            r = 0; bag = DLRM_DEAD_BAG;
        }
        const long long pos = base + i;
        keys[pos] = ((KT)t << row_bits) | (KT)r;
        if (vals) vals[pos] = (unsigned)pos;      // (the segmented sorter's first round takes the position itself: vals == nullptr)
        bag_of[pos] = bag;
    }
}

// A sorted entry is SINGLE when no other lookup of the batch names its (table, row) and the lookup itself is in range — the rows the fused
// backward (dlrm_interact_bwd_gather_sgd) updates itself and dlrm_emb_bwd_sgd_presorted skips.  ONE rule, used by both sides.
template <typename KT>
__device__ __forceinline__ bool sorted_entry_is_single(KT k, bool has_prev, KT prev, bool has_next, KT next, unsigned bag) {
    return (!has_prev || prev != k) && (!has_next || next != k) && bag != DLRM_DEAD_BAG;
}

// single_mask[bag] |= 1 << table for every single entry of the sorted list (mask zeroed by expand_kernel; one lookup per bag: a bit names a lookup)
template <typename KT>
__global__ __launch_bounds__(256) void single_mask_kernel(long long L, int row_bits, const KT* __restrict__ keys, const unsigned* __restrict__ vals,
                                                          const unsigned* __restrict__ bag_of, unsigned* __restrict__ mask) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= L) return;
    const KT k = keys[j];
    const bool hp = j > 0, hn = j + 1 < L;
    const KT pk = hp ? keys[j - 1] : k, nk = hn ? keys[j + 1] : k;
    if (pk == k && hp) return;
    if (nk == k && hn) return;
    const unsigned bag = bag_of[vals[j]];
    if (sorted_entry_is_single<KT>(k, hp, pk, hn, nk, bag)) atomicOr(mask + bag, 1u << (unsigned)(k >> row_bits));
}

static int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
static int bits_for(long long n) { int b = 0; while (((long long)1 << b) < n) ++b; return b < 1 ? 1 : b; }
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t keys_in, keys_out, vals_in, vals_out, bag_of, temp, temp_bytes, total;
    // the segmented sorter of seg_sort.h (own = true: it handles this group; the rocPRIM temp then stays unused)
    size_t keys_tmp, vals_tmp, hist, binbase, gtot, bsum;
    bool own;
    SegPlan plan;
};

// Which sorter takes a launch group.  DESIGN DECISION (round 6): the library's own segmented sorter (seg_sort.h) for table
// segments of up to 262144 lookups — every one-lookup-per-bag batch, i.e. the headline — and the vendor's general radix sort (rocPRIM
// onesweep) for longer segments (the multi-hot MLPerf-v2 batch: measured 773 vs 1146 us per 14 M-lookup sort, profiles/round4/
// sort_long_segments.md).  Tuning builds only: env DLRM_SORT = "rocprim" (the general sorter everywhere) | "own" (the segmented sorter for
// long segments too: correct, tested through tools/, slower).
static int seg_sort_mode() {            // 0 rocprim, 1 default, 2 own everywhere
    static const int v = [] {
        const char* e = DLRM_TUNE_ENV_STR("DLRM_SORT");
        return (e && strcmp(e, "rocprim") == 0) ? 0 : (e && strcmp(e, "own") == 0) ? 2 : 1;
    }();
    return v;
}
static bool seg_sort_enabled() { return seg_sort_mode() != 0; }

template <typename KT>
static hipError_t sort_temp_bytes(size_t L, int bits, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs<rocprim::default_config, const KT*, KT*, const unsigned*, unsigned*>(
        nullptr, *bytes, nullptr, nullptr, nullptr, nullptr, L, 0, bits, (hipStream_t)0, false);
}

// n / nnz / rows: the tables of the launch group (in launch order) — given, the group is planned for the segmented sorter
static int make_layout(size_t L, bool wide, int bits, Layout* lo, int n = 0, const int64_t* nnz = nullptr, const int64_t* rows = nullptr) {
    const size_t ksz = wide ? 8 : 4;
    size_t o = 0;
    lo->own = false;
    if (n > 0 && n <= DLRM_MAX_TABLES_PER_LAUNCH && nnz && rows && seg_sort_enabled()) {
        long long nz[DLRM_MAX_TABLES_PER_LAUNCH], rw[DLRM_MAX_TABLES_PER_LAUNCH];
        for (int k = 0; k < n; ++k) { nz[k] = (long long)nnz[k]; rw[k] = (long long)rows[k]; }
        lo->own = seg_plan(n, nz, rw, &lo->plan, seg_sort_mode() == 2);
    }
    lo->keys_in = o;  o += align256(L * ksz);
    lo->keys_out = o; o += align256(L * ksz);
    lo->vals_in = o;  o += align256(L * 4);
    lo->vals_out = o; o += align256(L * 4);
    lo->bag_of = o;   o += align256(L * 4);
    lo->temp = o; lo->temp_bytes = 0;
    lo->keys_tmp = lo->vals_tmp = lo->hist = lo->binbase = lo->gtot = lo->bsum = 0;
    if (lo->own) {
        // (the general sorter's temporary storage is neither sized nor reserved when the segmented sorter takes the group)
        lo->keys_tmp = o; o += align256(L * ksz);
        lo->vals_tmp = o; o += align256(L * 4);
        lo->hist = o;     o += align256(lo->plan.hist_words * 4);
        lo->binbase = o;  o += align256(lo->plan.bin_words * 4);
        lo->gtot = o;     o += align256(lo->plan.gtot_words * 4);
        lo->bsum = o;     o += align256(lo->plan.bsum_words * 4);
    } else {
        hipError_t e = wide ? sort_temp_bytes<unsigned long long>(L, bits, &lo->temp_bytes)
                            : sort_temp_bytes<unsigned>(L, bits, &lo->temp_bytes);
        if (e != hipSuccess) return (int)e;
        o += align256(lo->temp_bytes);
    }
    lo->total = o;
    return 0;
}


// Fills the by-value kernel arguments for tables ids[0..n), expands every lookup to (key = table << row_bits | row,
// value = global lookup position, bag_of[position] = bag) and radix-sorts the pairs by key (stable: equal rows keep
// input order).  Results: keys_out / vals_out / bag_of inside `ws` as laid out by `lo`.
template <typename KT>
static int expand_and_sort(int n, const int* ids, int64_t B, void* const* weight_host, const int64_t* rows_host,
                           const void* const* indices_host, const void* const* offsets_host, const int64_t* nnz_host,
                           const void* const* psw_host, int idx_bits, char* ws, const Layout& lo, size_t L, int row_bits,
                           int key_bits, hipStream_t st, SortedArgs* sa_out, int64_t* err, unsigned* single_mask = nullptr) {
    EmbArgs a;
    a.err = (long long*)err; a.pred.flag = nullptr; a.pred.nonzero = 0;
    SortedArgs& sa = *sa_out;
    long long base = 0;
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) {
        const int t = ids[k < n ? k : 0];
        a.w[k] = (float*)weight_host[t]; a.idx[k] = indices_host[t]; a.off[k] = offsets_host[t];
        a.psw[k] = psw_host ? (const float*)psw_host[t] : nullptr;
        a.nnz[k] = k < n ? nnz_host[t] : 0; a.rows[k] = rows_host[t]; a.slot[k] = t;
        sa.w[k] = a.w[k]; sa.psw[k] = a.psw[k]; sa.slot[k] = t; sa.base[k] = base;
        if (k < n) base += nnz_host[t];
    }
    KT* keys_in = (KT*)(ws + lo.keys_in);
    KT* keys_out = (KT*)(ws + lo.keys_out);
    unsigned* vals_in = (unsigned*)(ws + lo.vals_in);
    unsigned* vals_out = (unsigned*)(ws + lo.vals_out);
    unsigned* bag_of = (unsigned*)(ws + lo.bag_of);
    dim3 block(256);
    if (single_mask && L != (size_t)n * (size_t)B) return DLRM_E_ARG;      // (a mask bit names ONE lookup of a bag)
    if (L > (size_t)2 * (size_t)n * (size_t)B) {
        // multi-hot: one thread per lookup (coalesced stores).  grid.x covers the largest table in one sweep, capped; smaller tables'
        // surplus blocks exit at once
        long long mx = 0;
        for (int k = 0; k < n; ++k) if (a.nnz[k] > mx) mx = a.nnz[k];
        long long gx = (mx + 255) / 256; if (gx > 8192) gx = 8192; if (gx < 1) gx = 1;
        dim3 grid((unsigned)gx, (unsigned)n, 1);
        if (idx_bits == 64)
            hipLaunchKernelGGL((expand_positions_kernel<long long, KT>), grid, block, 0, st, a, sa, (long long)B, row_bits, keys_in, lo.own ? nullptr : vals_in, bag_of);
        else
            hipLaunchKernelGGL((expand_positions_kernel<int, KT>), grid, block, 0, st, a, sa, (long long)B, row_bits, keys_in, lo.own ? nullptr : vals_in, bag_of);
    } else {
        dim3 grid((unsigned)((B + 255) / 256), (unsigned)n, 1);
        if (idx_bits == 64)
            hipLaunchKernelGGL((expand_kernel<long long, KT>), grid, block, 0, st, a, sa, (long long)B, row_bits, keys_in, lo.own ? nullptr : vals_in, bag_of, single_mask);
        else
            hipLaunchKernelGGL((expand_kernel<int, KT>), grid, block, 0, st, a, sa, (long long)B, row_bits, keys_in, lo.own ? nullptr : vals_in, bag_of, single_mask);
    }
    DLRM_LAUNCH_CHECK();
    if (lo.own) {       // table-major segments, per-table digit counts: seg_sort.h (graph-replayable: plain kernels, no memsets)
        const int rc = seg_sort_run<KT>(lo.plan, (const KT*)keys_in, (KT*)(ws + lo.keys_tmp), keys_out, (unsigned*)(ws + lo.vals_tmp), vals_out,
                                        (unsigned*)(ws + lo.hist), (unsigned*)(ws + lo.binbase), (unsigned*)(ws + lo.gtot), (unsigned*)(ws + lo.bsum), st);
        if (rc) return rc;
    } else {
        size_t tb = lo.temp_bytes;
        hipError_t e = rocprim::radix_sort_pairs(ws + lo.temp, tb, (const KT*)keys_in, keys_out, (const unsigned*)vals_in,
                                                 vals_out, L, 0, key_bits, st, false);
        if (e != hipSuccess) return (int)e;
    }
    if (single_mask) {
        hipLaunchKernelGGL((single_mask_kernel<KT>), dim3((unsigned)((L + 255) / 256)), block, 0, st, (long long)L, row_bits, (const KT*)keys_out,
                           (const unsigned*)vals_out, (const unsigned*)bag_of, single_mask);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace
