// emb_sorted.hip — fused EmbeddingBag backward + sparse SGD WITHOUT one atomic per element:
// lookups are radix-sorted by (table, row); every group of lanes then owns a short run of the sorted
// list and applies each row's update with a plain read-modify-write.
//
// Why: gfx950 has 8 XCDs with private L2s, so device-scope fp32 atomics execute memory-side.  The
// direct-atomic kernel (emb.hip) issues 128 of them per looked-up row and measured ~1 TB/s of
// algorithmic traffic on Criteo-Terabyte shapes; sorted runs need an atomic only where a row's run
// straddles two chunks.
//
// Reference semantics replaced: EmbeddingBagBackward (sparse COO) + torch.optim.SGD.step
// (dlrm_s_pytorch.py:1613,1620).  Within a run (stable sort => input order) the update is the same
// per-lookup chain W = fma(-lr, g, W) the reference executes, so rows whose run lies inside one chunk
// — every row of a large table in practice — are bit-identical to the reference.
#include "sorted_common.h"

namespace {

// Each group of LPB lanes owns Q*C consecutive entries of the sorted list and streams through them C at a time
// (C gradient rows + the table rows of the runs that START in the chunk are in flight together).  A run is carried
// across chunks in registers, so a row costs one table-row read and one write however many lookups hit it, and only
// the runs that cross a GROUP boundary (every Q*C = 64 entries) need atomics: the hot rows of tiny tables take
// 1/64 of an atomic row-add per lookup instead of 1/8 (measured: the nine < 1k-row Criteo tables took 913 of the
// 1340 us of the whole update before).
// SK (dlrm_emb_bwd_sgd_presorted with skip_singles): the SINGLE entries of the sorted list (sorted_entry_is_single: a row one lookup of
// the batch names) were already updated by the fused backward (dlrm_interact_bwd_gather_sgd) when `singles` holds — the same launch
// predicate that kernel ran under — and are skipped here: no gradient row, no table row, no store.
// PF (the presorted instance): the group fetches the (key, position, bag) of all its Q*C entries ONCE — lane l those of entries l, l + LPB, ...
// (coalesced; the bags one gather) — and a chunk takes them from the owning lane by a cross-lane read, so that its row loads depend on registers
// only: without it every chunk walks keys / positions -> bag -> gradient row, three dependent memory latencies, sixteen times per group
// (what adagrad_groups_kernel does since round 4).  Same entries, same order, same arithmetic.
__device__ __forceinline__ unsigned su_bcast(unsigned v, int src) { return (unsigned)__shfl((int)v, src, 64); }
__device__ __forceinline__ unsigned long long su_bcast(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, 64), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
template <typename T, int N> __device__ __forceinline__ T su_pick(const T (&a)[N], int slot) {
    T v = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = slot == i ? a[i] : v;
    return v;
}

template <int VEC, int LPB, int NCH, typename KT, int C, int Q, bool SK = false, bool PF = false>
__global__ __launch_bounds__(256) void sorted_update_kernel(SortedArgs sa, long long L, int D, int row_bits,
                                                            const KT* __restrict__ keys,
                                                            const unsigned* __restrict__ vals,
                                                            const unsigned* __restrict__ bag_of,
                                                            const float* __restrict__ dout, long long dout_ld,
                                                            DlrmStep neg_lr_, DlrmPred singles = DlrmPred{nullptr, 0}) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    const bool singles_done = SK && !singles.skip();
    using VT = typename Vec<VEC>::T;
    __shared__ long long s_w[DLRM_MAX_TABLES_PER_LAUNCH];
    __shared__ long long s_psw[DLRM_MAX_TABLES_PER_LAUNCH];
    __shared__ long long s_base[DLRM_MAX_TABLES_PER_LAUNCH];
    __shared__ int s_slot[DLRM_MAX_TABLES_PER_LAUNCH];
#pragma unroll
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k)
        if (threadIdx.x == k) {
            s_w[k] = (long long)sa.w[k]; s_psw[k] = (long long)sa.psw[k]; s_base[k] = sa.base[k]; s_slot[k] = sa.slot[k];
        }
    __syncthreads();

    constexpr int GPB = 256 / LPB;
    const int g = threadIdx.x / LPB, lig = threadIdx.x % LPB;
    const long long g0 = ((long long)blockIdx.x * GPB + g) * (C * Q);
    if (g0 >= L) return;
    const long long g_end = g0 + C * Q;                 // first entry of the next group
    const KT row_mask = (((KT)1) << row_bits) - 1;
    const KT none = (KT)~(KT)0;                         // filler for dead entries only; never compared as a key

    const bool has_prev = g0 > 0, has_end = g_end < L;
    const KT prev_key = has_prev ? keys[g0 - 1] : none;
    const KT end_key = has_end ? keys[g_end] : none;

    constexpr int NS = PF ? (C * Q) / LPB : 1;
    static_assert(!PF || ((C * Q) % LPB == 0 && LPB % C == 0 && LPB <= 64), "PF: chunks must not straddle the lanes' slots");
    const int lane0 = ((int)(threadIdx.x & 63) / LPB) * LPB;          // first lane of this group inside its wave
    KT e_key[NS]; unsigned e_pos[NS], e_bag[NS];
    if constexpr (PF) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const long long e = g0 + s_ * LPB + lig;
            e_key[s_] = e < L ? keys[e] : none;
            e_pos[s_] = e < L ? vals[e] : 0u;
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) e_bag[s_] = (g0 + s_ * LPB + lig < L) ? bag_of[e_pos[s_]] : 0u;
    }

    // the run being accumulated
    KT run_key = none;
    bool have_run = false;
    VT acc[NCH];
    float* run_row = nullptr;
    bool run_atomic = false;
#pragma unroll
    for (int c = 0; c < NCH; ++c) v_zero(acc[c]);

    auto flush = [&]() {
        if (run_row == nullptr) return;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (c * LPB + lig) * VEC;
            if (col < D) {
                if (run_atomic) v_gatomic_add(run_row + col, acc[c]);
                else v_gstore(run_row + col, acc[c]);
            }
        }
    };

    for (int it = 0; it < Q; ++it) {
        const long long c0 = g0 + (long long)it * C;
        if (c0 >= L) break;
        KT k[C];
        unsigned pos[C];
        bool live[C], starts[C], sk[C];
        unsigned bagv[C];
        const int slot = PF ? (it * C) / LPB : 0, src0 = lane0 + (it * C) % LPB;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            live[j] = c0 + j < L;
            if constexpr (PF) {
                k[j] = su_bcast(su_pick<KT, NS>(e_key, slot), src0 + j);
                pos[j] = su_bcast(su_pick<unsigned, NS>(e_pos, slot), src0 + j);
                bagv[j] = su_bcast(su_pick<unsigned, NS>(e_bag, slot), src0 + j);
            } else {
                k[j] = live[j] ? keys[c0 + j] : none;
                pos[j] = live[j] ? vals[c0 + j] : 0u;
            }
            sk[j] = false;
        }
#pragma unroll
        for (int j = 0; j < C; ++j) starts[j] = live[j] && (j == 0 ? (!have_run || k[0] != run_key) : (k[j] != k[j - 1]));
        // (SK) the key on either side of the chunk: the entry before it is the previous chunk's last one (run_key) or the previous group's
        bool c_hp = false, c_hn = false; KT c_pk = none, c_nk = none;
        if constexpr (SK) {
            c_hp = c0 > 0; c_pk = it == 0 ? prev_key : run_key;
            c_hn = c0 + C < L;
            if constexpr (PF) {
                // the entry behind the chunk: the next lane's, the next slot's first, or the next group's first (end_key)
                const int nx = (it * C) % LPB + C;
                const KT same = su_bcast(su_pick<KT, NS>(e_key, slot), lane0 + (nx < LPB ? nx : 0));
                const KT next = su_bcast(su_pick<KT, NS>(e_key, slot + 1 < NS ? slot + 1 : slot), lane0);
                c_nk = nx < LPB ? same : (slot + 1 < NS ? next : end_key);
            } else {
                c_nk = c_hn ? keys[c0 + C] : none;
            }
        }

        // everything this chunk needs from memory, issued up front: C gradient rows, and the table row of every run
        // that starts here and will be written with a plain store (= starts inside the group and ends inside it)
        VT gr[C][NCH], wr[C][NCH];
        float sc[C];
        float* wrow[C];
        bool atom[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int t = live[j] ? (int)(k[j] >> row_bits) : 0;
            const long long row = live[j] ? (long long)(k[j] & row_mask) : 0;
            unsigned bag;
            if constexpr (PF) bag = live[j] ? bagv[j] : 0u;
            else bag = live[j] ? bag_of[pos[j]] : 0u;
            if constexpr (SK) {
                const bool hp = j == 0 ? c_hp : true, hn = j == C - 1 ? c_hn : (c0 + j + 1 < L);
                const KT pk = j == 0 ? c_pk : k[j == 0 ? 0 : j - 1], nk = j == C - 1 ? c_nk : k[j == C - 1 ? j : j + 1];
                sk[j] = singles_done && live[j] && sorted_entry_is_single<KT>(k[j], hp, pk, hn, nk, bag);
            }
            const bool dead = bag == DLRM_DEAD_BAG;            // out-of-range lookup (expand_kernel): zero gradient
            if (dead) bag = 0u;
            const float* psw = (const float*)s_psw[t];
            sc[j] = dead ? 0.f : ((live[j] && psw) ? neg_lr * *(const sc_gfloat*)(psw + ((long long)pos[j] - s_base[t])) : neg_lr);
            wrow[j] = (float*)s_w[t] + row * D;
            // a run needs atomics iff it began before this group or continues into the next one
            atom[j] = (it == 0 && j == 0 && has_prev && k[j] == prev_key) || (has_end && k[j] == end_key);
            const float* grow = dout + (long long)bag * dout_ld + (long long)s_slot[t] * D;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (c * LPB + lig) * VEC;
                v_zero(gr[j][c]); v_zero(wr[j][c]);
                if (live[j] && col < D && !sk[j]) {
                    if (!dead) gr[j][c] = *(const VT*)(grow + col);
                    if (starts[j] && !atom[j]) v_gload(wr[j][c], wrow[j] + col);
                }
            }
        }

        // walk the chunk: W = fma(-lr*psw, g, W) per lookup, in (stable-sorted = input) order
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (!live[j]) break;
            if (SK && sk[j]) {          // nothing to do for this entry; it ends the run before it and is not part of the one after it
                flush();
                run_key = k[j]; run_row = nullptr; have_run = true;
                continue;
            }
            if (starts[j]) {
                flush();
                run_key = k[j]; run_row = wrow[j]; run_atomic = atom[j]; have_run = true;
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] = wr[j][c];     // zero for atomic runs (not loaded)
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) v_fma(acc[c], sc[j], gr[j][c]);
        }
    }
    flush();
}

#ifndef DLRM_SORTED_PF
#define DLRM_SORTED_PF true
#endif
template <typename KT>
static int run_sorted(int n, const int* ids, int64_t B, int D, void* const* weight_host, const int64_t* rows_host,
                      const void* const* indices_host, const void* const* offsets_host, const int64_t* nnz_host,
                      const void* const* psw_host, int idx_bits, const float* dout, int64_t dout_ld, DlrmStep neg_lr,
                      char* ws, const Layout& lo, size_t L, int row_bits, int key_bits, bool vec_ok, hipStream_t st,
                      int64_t* err) {
    SortedArgs sa;
    int rc0 = expand_and_sort<KT>(n, ids, B, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, idx_bits,
                                  ws, lo, L, row_bits, key_bits, st, &sa, err);
    if (rc0) return rc0;
    const KT* keys_out = (const KT*)(ws + lo.keys_out);
    const unsigned* vals_out = (const unsigned*)(ws + lo.vals_out);
    const unsigned* bag_of = (const unsigned*)(ws + lo.bag_of);
    const dim3 block(256);

    int vec = (vec_ok && D % 4 == 0) ? 4 : 1;
    const int units = vec == 4 ? D / 4 : D;
    int lpb = pow2ceil(units); if (lpb < 4) lpb = 4; if (lpb > 64) lpb = 64;
    int nch = (units + lpb - 1) / lpb; if (nch == 3) nch = 4;
    if (nch > 4) return DLRM_E_RANGE;
    const int gpb = 256 / lpb;
    const int cc = 8 / nch;                          // entries per chunk: 8 float4 pairs of registers per lane
    constexpr int kQ = 8;                            // chunks per group: 64 / nch consecutive entries per group
    const size_t per_wg = (size_t)gpb * cc * kQ;
    dim3 ugrid((unsigned)((L + per_wg - 1) / per_wg), 1, 1);
    // entries per chunk (x NCH registers), 64 entries per group either way; tuning builds: env DLRM_SORTED_C = 8 | 4 (default) | 2.  Measured on
                                // one box (profiles/r03/ceilings.md): 0.626 / 0.524 / 0.538 ms for the whole update at Criteo-Terabyte shapes — occupancy beats
                                // per-wave loads in flight
    static const int cdiv = DLRM_TUNE_ENV("DLRM_SORTED_C", 4);
#define SU_ARGS ugrid, block, 0, st, sa, (long long)L, D, row_bits, (const KT*)keys_out, (const unsigned*)vals_out, (const unsigned*)bag_of, dout, (long long)dout_ld, neg_lr
#define SU(V, LP, NC)                                                                                              \
    do {                                                                                                           \
        if (cdiv == 2 && NC == 1)      hipLaunchKernelGGL((sorted_update_kernel<V, LP, NC, KT, 2, 32>), SU_ARGS);   \
        else if (cdiv == 4 && NC == 1) hipLaunchKernelGGL((sorted_update_kernel<V, LP, NC, KT, 4, 16, false, (DLRM_SORTED_PF && LP >= 16)>), SU_ARGS);   \
        else                           hipLaunchKernelGGL((sorted_update_kernel<V, LP, NC, KT, 8 / NC, kQ>), SU_ARGS); \
    } while (0)
    const int key = vec * 10000 + lpb * 10 + nch;
    switch (key) {
        case 40041: SU(4, 4, 1); break;   case 40081: SU(4, 8, 1); break;   case 40161: SU(4, 16, 1); break;
        case 40321: SU(4, 32, 1); break;  case 40641: SU(4, 64, 1); break;  case 40642: SU(4, 64, 2); break;
        case 40644: SU(4, 64, 4); break;
        case 10041: SU(1, 4, 1); break;   case 10081: SU(1, 8, 1); break;   case 10161: SU(1, 16, 1); break;
        case 10321: SU(1, 32, 1); break;  case 10641: SU(1, 64, 1); break;  case 10642: SU(1, 64, 2); break;
        case 10644: SU(1, 64, 4); break;
        default: return DLRM_E_RANGE;
    }
#undef SU
#undef SU_ARGS
    DLRM_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// workspace query for DLRM_UPD_SORTED (0 for the other modes)
extern "C" int64_t dlrm_emb_bwd_workspace_bytes(int T, const int64_t* nnz_host, const int64_t* rows_host) {
    if (T <= 0 || !nnz_host || !rows_host) return 0;
    size_t worst = 0;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        size_t L = 0; long long max_rows = 1;
        for (int k = 0; k < n; ++k) { L += (size_t)nnz_host[t0 + k]; if (rows_host[t0 + k] > max_rows) max_rows = rows_host[t0 + k]; }
        if (L == 0) continue;
        const int row_bits = bits_for(max_rows), key_bits = row_bits + bits_for(n);
        Layout lo;
        if (make_layout(L, key_bits > 32, key_bits, &lo, n, nnz_host + t0, rows_host + t0) != 0) return -1;
        if (lo.total > worst) worst = lo.total;
    }
    return (int64_t)worst;
}

extern "C" int dlrm_emb_sort_kind(int T, const int64_t* nnz_host, const int64_t* rows_host) {
    if (T <= 0 || !nnz_host || !rows_host || !seg_sort_enabled()) return 0;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        long long nz[DLRM_MAX_TABLES_PER_LAUNCH], rw[DLRM_MAX_TABLES_PER_LAUNCH];
        for (int k = 0; k < n; ++k) { nz[k] = (long long)nnz_host[t0 + k]; rw[k] = (long long)rows_host[t0 + k]; }
        SegPlan plan;
        if (!seg_plan(n, nz, rw, &plan, seg_sort_mode() == 2)) return 0;
    }
    return 1;
}

namespace {
template <typename KT>
__global__ __launch_bounds__(256) void widen_keys_kernel(long long L, const KT* __restrict__ k, unsigned long long* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long long)gridDim.x * 256) out[i] = (unsigned long long)k[i];
}
}  // namespace

// The sort of the sort-based updates on its own (what dlrm_emb_bwd_sgd(DLRM_UPD_SORTED) and dlrm_emb_bwd_rowwise_adagrad run first):
// positions_out[j] = global lookup position (table-major) of the j-th entry in (table, row) order, equal rows in input order;
// keys_out[j] = table << row_bits | row of that entry (row_bits = bits of the largest table); bag_out[p] = bag of POSITION p
// (0xFFFFFFFF for a skipped out-of-range lookup).  One launch group only (T <= 32).  Used by the tests to check the sorter itself.
extern "C" int dlrm_emb_sort_lookups(int T, int64_t B, const int64_t* rows_host, const void* const* indices_host,
                                     const void* const* offsets_host, const int64_t* nnz_host, int idx_bits, void* workspace,
                                     int64_t workspace_bytes, uint32_t* positions_out, uint64_t* keys_out, uint32_t* bag_out,
                                     int* row_bits_out, int64_t* err, void* stream) {
    if (T <= 0 || T > DLRM_MAX_TABLES_PER_LAUNCH || B <= 0 || !rows_host || !indices_host || !offsets_host || !nnz_host || !positions_out ||
        !keys_out)
        return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    int ids[DLRM_MAX_TABLES_PER_LAUNCH];
    void* wfake[DLRM_MAX_TABLES_PER_LAUNCH];
    size_t L = 0; long long max_rows = 1;
    for (int k = 0; k < T; ++k) { ids[k] = k; wfake[k] = nullptr; L += (size_t)nnz_host[k]; if (rows_host[k] > max_rows) max_rows = rows_host[k]; }
    if (L == 0 || L >= ((size_t)1 << 32)) return DLRM_E_RANGE;
    const int row_bits = bits_for(max_rows), key_bits = row_bits + bits_for(T);
    const bool wide = key_bits > 32;
    if (row_bits_out) *row_bits_out = row_bits;
    Layout lo;
    int rc = make_layout(L, wide, key_bits, &lo, T, nnz_host, rows_host);
    if (rc) return rc;
    if (!workspace || (size_t)workspace_bytes < lo.total) return DLRM_E_ARG;
    char* ws = (char*)workspace;
    SortedArgs sa;
    rc = wide ? expand_and_sort<unsigned long long>(T, ids, B, wfake, rows_host, indices_host, offsets_host, nnz_host, nullptr, idx_bits, ws, lo,
                                                    L, row_bits, key_bits, st, &sa, err)
              : expand_and_sort<unsigned>(T, ids, B, wfake, rows_host, indices_host, offsets_host, nnz_host, nullptr, idx_bits, ws, lo, L, row_bits,
                                          key_bits, st, &sa, err);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(positions_out, ws + lo.vals_out, L * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && bag_out) e = hipMemcpyAsync(bag_out, ws + lo.bag_of, L * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    const unsigned nb = (unsigned)((L + 255) / 256 > 2048 ? 2048 : (L + 255) / 256);
    if (wide) hipLaunchKernelGGL((widen_keys_kernel<unsigned long long>), dim3(nb), dim3(256), 0, st, (long long)L, (const unsigned long long*)(ws + lo.keys_out), (unsigned long long*)keys_out);
    else      hipLaunchKernelGGL((widen_keys_kernel<unsigned>), dim3(nb), dim3(256), 0, st, (long long)L, (const unsigned*)(ws + lo.keys_out), (unsigned long long*)keys_out);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// ---- ABI 17: the sort in front of the fused backward, the update behind it ---------------------------------------------------------------
// dlrm_emb_presort = the first half of dlrm_emb_bwd_sgd(DLRM_UPD_SORTED) (expand + sort into `workspace`) + the per-bag mask of SINGLE lookups;
// dlrm_emb_bwd_sgd_presorted = its second half from that workspace.  One launch group (T <= 32), one lookup per bag position (nnz[t] == B).
namespace {
struct PresortGroup { size_t L; int row_bits, key_bits; bool wide; Layout lo; };
static int presort_group(int T, int64_t B, const int64_t* rows_host, const int64_t* nnz_host, PresortGroup* g) {
    if (T <= 0 || T > DLRM_MAX_TABLES_PER_LAUNCH || B <= 0 || !rows_host || !nnz_host) return DLRM_E_ARG;
    size_t L = 0; long long max_rows = 1;
    for (int k = 0; k < T; ++k) {
        if (nnz_host[k] != B || rows_host[k] <= 0) return DLRM_E_ARG;
        L += (size_t)nnz_host[k];
        if (rows_host[k] > max_rows) max_rows = rows_host[k];
    }
    if (L >= ((size_t)1 << 32)) return DLRM_E_RANGE;
    g->L = L; g->row_bits = bits_for(max_rows); g->key_bits = g->row_bits + bits_for(T); g->wide = g->key_bits > 32;
    return make_layout(L, g->wide, g->key_bits, &g->lo, T, nnz_host, rows_host);
}
}  // namespace

extern "C" int dlrm_emb_presort(int T, int64_t B, const int64_t* rows_host, const void* const* indices_host, const void* const* offsets_host,
                                const int64_t* nnz_host, int idx_bits, void* workspace, int64_t workspace_bytes, uint32_t* single_mask,
                                int64_t* err, void* stream) {
    if (!indices_host || !offsets_host || !single_mask) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    PresortGroup g;
    int rc = presort_group(T, B, rows_host, nnz_host, &g);
    if (rc) return rc;
    if (!workspace || (size_t)workspace_bytes < g.lo.total) return DLRM_E_ARG;
    int ids[DLRM_MAX_TABLES_PER_LAUNCH];
    void* wfake[DLRM_MAX_TABLES_PER_LAUNCH];
    for (int k = 0; k < T; ++k) { ids[k] = k; wfake[k] = nullptr; }
    SortedArgs sa;
    hipStream_t st = (hipStream_t)stream;
    return g.wide ? expand_and_sort<unsigned long long>(T, ids, B, wfake, rows_host, indices_host, offsets_host, nnz_host, nullptr, idx_bits,
                                                        (char*)workspace, g.lo, g.L, g.row_bits, g.key_bits, st, &sa, err, (unsigned*)single_mask)
                  : expand_and_sort<unsigned>(T, ids, B, wfake, rows_host, indices_host, offsets_host, nnz_host, nullptr, idx_bits,
                                              (char*)workspace, g.lo, g.L, g.row_bits, g.key_bits, st, &sa, err, (unsigned*)single_mask);
}

#ifndef DLRM_PRESORTED_PF
#define DLRM_PRESORTED_PF true
#endif
namespace {
template <typename KT>
static int run_presorted(int T, int D, void* const* weight_host, const int64_t* nnz_host, const PresortGroup& g, const char* ws, const float* dout,
                         int64_t dout_ld, DlrmStep neg_lr, bool skip_singles, DlrmPred singles, hipStream_t st) {
    SortedArgs sa;
    long long base = 0;
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) {
        const int t = k < T ? k : 0;
        sa.w[k] = (float*)weight_host[t]; sa.psw[k] = nullptr; sa.slot[k] = t; sa.base[k] = base;
        if (k < T) base += nnz_host[t];
    }
    // D = 128: 32 lanes x float4 per row, 4 entries per chunk, 16 chunks per group (the instance dlrm_emb_bwd_sgd runs at this width)
    constexpr int LPB = 32, CC = 4, QQ = 16;
    const size_t per_wg = (size_t)(256 / LPB) * CC * QQ;
    const dim3 ugrid((unsigned)((g.L + per_wg - 1) / per_wg), 1, 1), block(256);
    const KT* keys = (const KT*)(ws + g.lo.keys_out);
    const unsigned* vals = (const unsigned*)(ws + g.lo.vals_out);
    const unsigned* bag_of = (const unsigned*)(ws + g.lo.bag_of);
    if (skip_singles)
        hipLaunchKernelGGL((sorted_update_kernel<4, LPB, 1, KT, CC, QQ, true, DLRM_PRESORTED_PF>), ugrid, block, 0, st, sa, (long long)g.L, D, g.row_bits, keys, vals,
                           bag_of, dout, (long long)dout_ld, neg_lr, singles);
    else
        hipLaunchKernelGGL((sorted_update_kernel<4, LPB, 1, KT, CC, QQ, false, DLRM_PRESORTED_PF>), ugrid, block, 0, st, sa, (long long)g.L, D, g.row_bits, keys, vals,
                           bag_of, dout, (long long)dout_ld, neg_lr, singles);
    DLRM_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int dlrm_emb_bwd_sgd_presorted(int T, int64_t B, int D, void* const* weight_host, const int64_t* rows_host, const int64_t* nnz_host,
                                          const float* dout, int64_t dout_ld, float lr, const float* lr_dev, const void* workspace,
                                          int64_t workspace_bytes, int skip_singles, const int32_t* pred_flag, int pred_nonzero, void* stream) {
    if (!weight_host || !dout || dout_ld < (int64_t)T * D) return DLRM_E_ARG;
    if (D != 128) return DLRM_E_MODE;                         // (the shapes of the fused lookup + interaction path)
    PresortGroup g;
    int rc = presort_group(T, B, rows_host, nnz_host, &g);
    if (rc) return rc;
    if (!workspace || (size_t)workspace_bytes < g.lo.total) return DLRM_E_ARG;
    bool vec_ok = dlrm_aligned16(dout) && (dout_ld % 4 == 0);
    for (int t = 0; t < T; ++t) { if (!weight_host[t]) return DLRM_E_ARG; vec_ok = vec_ok && dlrm_aligned16(weight_host[t]); }
    if (!vec_ok) return DLRM_E_MODE;
    const DlrmPred singles{(const int*)pred_flag, pred_nonzero};
    return g.wide ? run_presorted<unsigned long long>(T, D, weight_host, nnz_host, g, (const char*)workspace, dout, dout_ld, dlrm_step_neg(lr, lr_dev),
                                                      skip_singles != 0, singles, (hipStream_t)stream)
                  : run_presorted<unsigned>(T, D, weight_host, nnz_host, g, (const char*)workspace, dout, dout_ld, dlrm_step_neg(lr, lr_dev),
                                            skip_singles != 0, singles, (hipStream_t)stream);
}

int dlrm_emb_bwd_sgd_sorted_impl(int T, int64_t B, int D, void* const* weight_host, const int64_t* rows_host,
                                 const void* const* indices_host, const void* const* offsets_host,
                                 const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                                 const float* dout, int64_t dout_ld, float lr, const float* lr_dev, void* workspace,
                                 int64_t workspace_bytes, int64_t* err, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    bool vec_ok = dlrm_aligned16(dout) && (dout_ld % 4 == 0);
    for (int t = 0; t < T; ++t) vec_ok = vec_ok && dlrm_aligned16(weight_host[t]);
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        int ids[DLRM_MAX_TABLES_PER_LAUNCH];
        size_t L = 0; long long max_rows = 1;
        for (int k = 0; k < n; ++k) {
            ids[k] = t0 + k; L += (size_t)nnz_host[t0 + k];
            if (rows_host[t0 + k] > max_rows) max_rows = rows_host[t0 + k];
        }
        if (L == 0) continue;
        if (L >= ((size_t)1 << 32)) {
            fprintf(stderr, "libdlrm_hip: dlrm_emb_bwd_sgd(sorted): more than 2^32 lookups in one table group\n");
            return DLRM_E_RANGE;
        }
        const int row_bits = bits_for(max_rows), key_bits = row_bits + bits_for(n);
        const bool wide = key_bits > 32;
        Layout lo;
        int rc = make_layout(L, wide, key_bits, &lo, n, nnz_host + t0, rows_host + t0);
        if (rc) return rc;
        if (!workspace || (size_t)workspace_bytes < lo.total) {
            fprintf(stderr, "libdlrm_hip: dlrm_emb_bwd_sgd(sorted): workspace too small (%lld < %zu bytes)\n",
                    (long long)workspace_bytes, lo.total);
            return DLRM_E_ARG;
        }
        rc = wide ? run_sorted<unsigned long long>(n, ids, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host,
                                                   psw_host, idx_bits, dout, dout_ld, dlrm_step_neg(lr, lr_dev), (char*)workspace, lo, L, row_bits,
                                                   key_bits, vec_ok, st, err)
                  : run_sorted<unsigned>(n, ids, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host,
                                         idx_bits, dout, dout_ld, dlrm_step_neg(lr, lr_dev), (char*)workspace, lo, L, row_bits, key_bits, vec_ok, st, err);
        if (rc) return rc;
    }
    return 0;
}
