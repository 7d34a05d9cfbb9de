// adagrad.hip — K4: fused EmbeddingBag backward + row-wise sparse Adagrad, and the dense Adagrad step.
//
// Reference replaced: autograd EmbeddingBagBackward (sparse COO gradient) followed by RWSAdagrad.step's sparse
// branch (optim/rwsadagrad.py:117-143): coalesce the gradient (duplicates of a row summed), then per touched row r
//     mom[r] += mean_d(g_r[d]^2);   W[r,:] -= clr * g_r / (sqrt(mom[r]) + eps)
// and its dense branch (:145-148) for the MLP parameters:  sum += g*g;  w -= clr * g / (sqrt(sum) + eps).
//
// The update is non-linear in g_r, so a row's complete gradient must exist before it is applied.  Lookups of all
// tables are radix-sorted by (table,row) (shared with the SGD path, sorted_common.h); a group of lanes then owns 64
// consecutive sorted entries:
//   pass 1  walks its entries, sums each run of equal keys in input order (stable sort) and applies the update for
//           every run that lies inside the group; the partial sums of a run that continues from the previous group
//           or into the next one go to an edge buffer (plain stores, no atomics);
//   pass 2  one lane group per run that crosses a group boundary (owner = the group where the run starts) finds the
//           run's end by binary search in the sorted keys, adds the edge partials in order and applies the update.
// Every sum has a fixed order: results are run-to-run deterministic (hot rows are re-associated per 64 lookups
// relative to the reference's strictly sequential coalesce).
#include <stdlib.h>
#include "sorted_common.h"

namespace {

constexpr int kG = 64;          // sorted entries per lane group
// kC = gradient rows in flight per lane group: template parameter (env DLRM_ADAGRAD_KC, default 4)

struct AdagradArgs {
    float* state[DLRM_MAX_TABLES_PER_LAUNCH];      // row-wise accumulator ("momentum") of each table, [rows]
};

__device__ __forceinline__ float4 v_add(const float4& a, const float4& b) {
    return make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w));
}
__device__ __forceinline__ float v_add(const float& a, const float& b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float4 v_scale(float s, const float4& v) {
    return make_float4(__fmul_rn(s, v.x), __fmul_rn(s, v.y), __fmul_rn(s, v.z), __fmul_rn(s, v.w));
}
__device__ __forceinline__ float v_scale(float s, const float& v) { return __fmul_rn(s, v); }
__device__ __forceinline__ float v_sq(const float4& v) { return (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }
__device__ __forceinline__ float v_sq(const float& v) { return v * v; }
__device__ __forceinline__ void v_step(float4& w, float nclr_over_denom, const float4& g) {
    w.x = __builtin_fmaf(nclr_over_denom, g.x, w.x); w.y = __builtin_fmaf(nclr_over_denom, g.y, w.y);
    w.z = __builtin_fmaf(nclr_over_denom, g.z, w.z); w.w = __builtin_fmaf(nclr_over_denom, g.w, w.w);
}

// mom[row] += mean(g^2); W[row,:] = fma(-clr, g / (sqrt(mom[row]) + eps), W[row,:]).  Called by all LPB lanes of a
// lane group together (same control flow), `acc` = the lane's columns of the row's coalesced gradient.
template <int VEC, int LPB, int NCH>
__device__ __forceinline__ void adagrad_apply(float* __restrict__ wrow, float* __restrict__ mom_r, int D, int lig,
                                              const typename Vec<VEC>::T (&acc)[NCH], float clr, float eps) {
    using VT = typename Vec<VEC>::T;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * LPB + lig) * VEC;
        if (col < D) sq += v_sq(acc[c]);
    }
#pragma unroll
    for (int o = LPB >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    // (row and accumulator through GLOBAL pointers: sorted_common.h v_gload — generic ones made these flat instructions)
    const float m = *(const sc_gfloat*)mom_r + sq / (float)D;              // every lane reads the old value before lane 0 stores the new one
    const float denom = sqrtf(m) + eps;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * LPB + lig) * VEC;
        if (col < D) {
            VT w;
            v_gload(w, wrow + col);
            VT q;
            if constexpr (VEC == 4) q = make_float4(acc[c].x / denom, acc[c].y / denom, acc[c].z / denom, acc[c].w / denom);
            else q = acc[c] / denom;
            if constexpr (VEC == 4) v_step(w, -clr, q); else w = __builtin_fmaf(-clr, q, w);
            v_gstore(wrow + col, w);
        }
    }
    if (lig == 0) *(sc_gfloat*)mom_r = m;
}

// Pass 1.  A lookup's gradient row is found through a chain of dependent loads — sorted key, sorted value (= lookup position), bag of
// that position, row of dOut — and rounds 1-3 walked that chain once per ENTRY: 64 entries x 4 memory latencies per lane group, with one
// 512-byte row in flight per group (1.85 TB/s for the 14 M lookups of the MLPerf-v2 batch: latency-bound, not HBM-bound; more rows per step
// only lengthened the chains: DLRM_ADAGRAD_KC 1 / 2 / 4 = 5.28 / 5.55 / 5.8 ms).  Now the GROUP resolves all of its 64 entries at once: lane l
// loads the keys and values of entries l, l + LPB, ... (coalesced), gathers their bags (and pooling weights) — two latencies for the whole
// group — and the walk then takes (key, bag, weight) of entry j from lane j % LPB by a cross-lane read: the row loads depend on registers
// only and kC of them are in flight per group.  Same entries, same order, same sums: results are bit-identical to the old walk.
// Measured (round 4): emb_bwd_adagrad of the MLPerf-v2 batch 4.85 -> 3.62 ms, of the Terabyte batch 0.685 -> 0.56 ms.  (Also requesting the table
// row and accumulator of a run when it STARTS, so that its end waits for nothing, changed nothing — 3.69-3.76 ms: the walk is at ~0.7 of the copy
// rate on its real traffic by then, the rest of the category is the sort.)
template <typename T> __device__ __forceinline__ T group_bcast(T v, int src_lane);
template <> __device__ __forceinline__ unsigned group_bcast<unsigned>(unsigned v, int src_lane) { return (unsigned)__shfl((int)v, src_lane, 64); }
template <> __device__ __forceinline__ unsigned long long group_bcast<unsigned long long>(unsigned long long v, int src_lane) {
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src_lane, 64), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src_lane, 64);
    return ((unsigned long long)hi << 32) | lo;
}
template <> __device__ __forceinline__ float group_bcast<float>(float v, int src_lane) { return __shfl(v, src_lane, 64); }

template <int VEC, int LPB, int NCH, typename KT, int kC>
__global__ __launch_bounds__(256) void adagrad_groups_kernel(SortedArgs sa, AdagradArgs aa, long long L, int D, int row_bits,
                                                             const KT* __restrict__ keys, const unsigned* __restrict__ vals,
                                                             const unsigned* __restrict__ bag_of,
                                                             const float* __restrict__ dout, long long dout_ld,
                                                             DlrmStep clr_, float eps, float* __restrict__ edge_first,
                                                             float* __restrict__ edge_last) {
    const float clr = clr_;              // (by value, or read from the device scalar: common.h DlrmStep)
    using VT = typename Vec<VEC>::T;
    constexpr int DP = NCH * LPB * VEC;                  // padded row length of the edge buffers
    constexpr int GPB = 256 / LPB;
    constexpr int NS = kG / LPB;                         // entries a lane resolves for its group (LPB is a power of two <= 64)
    static_assert(kG % LPB == 0 && LPB % kC == 0, "group geometry");
    // per-table arguments are indexed by a lane-dependent table id: staged in LDS (a by-value kernel argument array indexed that way is
    // copied to scratch memory by the compiler)
    __shared__ long long s_w[DLRM_MAX_TABLES_PER_LAUNCH], s_state[DLRM_MAX_TABLES_PER_LAUNCH], s_psw[DLRM_MAX_TABLES_PER_LAUNCH],
        s_base[DLRM_MAX_TABLES_PER_LAUNCH];
    __shared__ int s_slot[DLRM_MAX_TABLES_PER_LAUNCH];
#pragma unroll
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k)
        if (threadIdx.x == k) {
            s_w[k] = (long long)sa.w[k]; s_state[k] = (long long)aa.state[k]; s_psw[k] = (long long)sa.psw[k]; s_base[k] = sa.base[k];
            s_slot[k] = sa.slot[k];
        }
    __syncthreads();
    const int g = threadIdx.x / LPB, lig = threadIdx.x % LPB;
    const int lane0 = (threadIdx.x & 63) & ~(LPB - 1);   // first lane of this group inside its wave
    const long long grp = (long long)blockIdx.x * GPB + g;
    const long long g0 = grp * kG;
    if (g0 >= L) return;
    const long long g_end = (g0 + kG < L) ? g0 + kG : L;
    const KT row_mask = (((KT)1) << row_bits) - 1;
    const bool cont_in = g0 > 0 && keys[g0 - 1] == keys[g0];
    const bool tail_cont = g0 + kG < L && keys[g0 + kG - 1] == keys[g0 + kG];

    // ---- resolve: entry g0 + s * LPB + lig for s = 0 .. NS-1
    KT e_key[NS];
    unsigned e_bag[NS];
    float e_sc[NS];                                      // pooling weight of the lookup; < 0 is never used as a marker: e_w says whether it applies
    bool e_w[NS];
    {
        unsigned pos[NS];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const long long e = g0 + s_ * LPB + lig;
            const bool live = e < g_end;
            e_key[s_] = live ? keys[e] : (KT)0;
            pos[s_] = live ? vals[e] : 0u;
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const bool live = g0 + s_ * LPB + lig < g_end;
            const int t = (int)(e_key[s_] >> row_bits);
            e_bag[s_] = live ? bag_of[pos[s_]] : DLRM_DEAD_BAG;
            const float* psw = (const float*)s_psw[t];
            e_w[s_] = live && psw != nullptr;
            e_sc[s_] = e_w[s_] ? psw[(long long)pos[s_] - s_base[t]] : 1.f;
        }
    }

    VT acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) v_zero(acc[c]);
    KT run_key = group_bcast<KT>(e_key[0], lane0);       // key of entry g0
    bool run_first = true, run_empty = true;

    auto finish = [&](bool is_last) {
        if (run_first && cont_in) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) *(VT*)(edge_first + grp * DP + (c * LPB + lig) * VEC) = acc[c];
        } else if (is_last && tail_cont) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) *(VT*)(edge_last + grp * DP + (c * LPB + lig) * VEC) = acc[c];
        } else {
            const int t = (int)(run_key >> row_bits);
            const long long row = (long long)(run_key & row_mask);
            adagrad_apply<VEC, LPB, NCH>((float*)s_w[t] + row * D, (float*)s_state[t] + row, D, lig, acc, clr, eps);
        }
    };

#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const long long s0 = g0 + s_ * LPB;
        if (s0 >= g_end) break;
        for (int j0 = 0; j0 < LPB; j0 += kC) {
            if (s0 + j0 >= g_end) break;
            KT k[kC];
            VT gr[kC][NCH];
            float sc[kC];
            bool live[kC], weighted[kC];
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                const int src = lane0 + j0 + j;
                live[j] = s0 + j0 + j < g_end;
                k[j] = group_bcast<KT>(e_key[s_], src);
                unsigned bag = group_bcast<unsigned>(e_bag[s_], src);
                sc[j] = group_bcast<float>(e_sc[s_], src);
                const bool w_ = __shfl((int)e_w[s_], src, 64) != 0;
                const int t = (int)(k[j] >> row_bits);
                const bool dead = bag == DLRM_DEAD_BAG;            // out-of-range lookup (expand_kernel): zero gradient
                if (dead) bag = 0u;
                weighted[j] = live[j] && w_ && !dead;
                const float* grow = dout + (long long)bag * dout_ld + (long long)s_slot[t] * D;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int col = (c * LPB + lig) * VEC;
                    v_zero(gr[j][c]);
                    if (live[j] && !dead && col < D) gr[j][c] = *(const VT*)(grow + col);
                }
            }
#pragma unroll
            for (int j = 0; j < kC; ++j) {
                if (!live[j]) break;
                if (k[j] != run_key) {
                    finish(false);
                    run_key = k[j]; run_first = false; run_empty = true;
                }

#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const VT v = weighted[j] ? v_scale(sc[j], gr[j][c]) : gr[j][c];
                    acc[c] = run_empty ? v : v_add(acc[c], v);
                }
                run_empty = false;
            }
        }
    }
    finish(true);
}

// one lane group per group index: acts only where a run starts in this group and continues into the next one
template <int VEC, int LPB, int NCH, typename KT>
__global__ __launch_bounds__(256) void adagrad_fixup_kernel(SortedArgs sa, AdagradArgs aa, long long L, int D, int row_bits,
                                                            const KT* __restrict__ keys, DlrmStep clr_, float eps,
                                                            const float* __restrict__ edge_first,
                                                            const float* __restrict__ edge_last) {
    const float clr = clr_;              // (by value, or read from the device scalar: common.h DlrmStep)
    using VT = typename Vec<VEC>::T;
    constexpr int DP = NCH * LPB * VEC;
    constexpr int GPB = 256 / LPB;
    const int g = threadIdx.x / LPB, lig = threadIdx.x % LPB;
    const long long grp = (long long)blockIdx.x * GPB + g;
    const long long g0 = grp * kG;
    if (g0 + kG >= L) return;                                   // no next group: nothing continues
    const KT key = keys[g0 + kG - 1];
    if (key != keys[g0 + kG]) return;                           // last run ends inside this group
    if (g0 > 0 && keys[g0 - 1] == key) return;                  // the run began before this group: not the owner
    // run end = first position with keys[pos] != key  (keys are sorted: binary search in (g0 + kG, L])
    long long lo = g0 + kG + 1, hi = L;                         // keys[lo - 1] == key
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (keys[mid] == key) lo = mid + 1; else hi = mid;
    }
    const long long last_grp = (lo - 1) / kG;                   // group holding the run's last entry (> grp)
    VT acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = *(const VT*)(edge_last + grp * DP + (c * LPB + lig) * VEC);
    long long j = grp + 1;
    for (; j + 7 <= last_grp; j += 8) {                         // 8 independent partial rows in flight
        VT p[8][NCH];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) p[u][c] = *(const VT*)(edge_first + (j + u) * DP + (c * LPB + lig) * VEC);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = v_add(acc[c], p[u][c]);
    }
    for (; j <= last_grp; ++j)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = v_add(acc[c], *(const VT*)(edge_first + j * DP + (c * LPB + lig) * VEC));
    const KT row_mask = (((KT)1) << row_bits) - 1;
    const int t = (int)(key >> row_bits);
    const long long row = (long long)(key & row_mask);
    adagrad_apply<VEC, LPB, NCH>(sa.w[t] + row * D, aa.state[t] + row, D, lig, acc, clr, eps);
}

struct Shape { int vec, lpb, nch, dp; };

static int pick(int D, bool vec_ok, Shape* s) {
    s->vec = (vec_ok && D % 4 == 0) ? 4 : 1;
    const int units = s->vec == 4 ? D / 4 : D;
    s->lpb = pow2ceil(units); if (s->lpb < 4) s->lpb = 4; if (s->lpb > 64) s->lpb = 64;
    s->nch = (units + s->lpb - 1) / s->lpb; if (s->nch == 3) s->nch = 4;
    if (s->nch > 4) return DLRM_E_RANGE;
    s->dp = s->nch * s->lpb * s->vec;
    return 0;
}

// edge buffers use the widest padded row any shape of this D can have (vector path: D rounded up; scalar path likewise)
static size_t edge_row_floats(int D) {
    Shape a, b;
    size_t m = 0;
    if (pick(D, true, &a) == 0) m = (size_t)a.dp;
    if (pick(D, false, &b) == 0 && (size_t)b.dp > m) m = (size_t)b.dp;
    return m;
}

struct AdaLayout { Layout sort; size_t edge_first, edge_last, total; };

static int ada_layout(size_t L, bool wide, int key_bits, int D, AdaLayout* lo, int n, const int64_t* nnz, const int64_t* rows) {
    int rc = make_layout(L, wide, key_bits, &lo->sort, n, nnz, rows);
    if (rc) return rc;
    const size_t groups = (L + kG - 1) / kG;
    const size_t row = edge_row_floats(D);
    if (row == 0) return DLRM_E_RANGE;
    size_t o = lo->sort.total;
    lo->edge_first = o; o += align256(groups * row * sizeof(float));
    lo->edge_last = o;  o += align256(groups * row * sizeof(float));
    lo->total = o;
    return 0;
}

template <typename KT>
static int run_adagrad(int n, const int* ids, int64_t B, int D, void* const* weight_host, void* const* state_host,
                       const int64_t* rows_host, const void* const* indices_host, const void* const* offsets_host,
                       const int64_t* nnz_host, const void* const* psw_host, int idx_bits, const float* dout, int64_t dout_ld,
                       DlrmStep clr, float eps, char* ws, const AdaLayout& lo, size_t L, int row_bits, int key_bits, bool vec_ok,
                       hipStream_t st, int64_t* err) {
    SortedArgs sa;
    int rc = expand_and_sort<KT>(n, ids, B, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, idx_bits, ws,
                                 lo.sort, L, row_bits, key_bits, st, &sa, err);
    if (rc) return rc;
    AdagradArgs aa;
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) aa.state[k] = (float*)state_host[ids[k < n ? k : 0]];
    const KT* keys = (const KT*)(ws + lo.sort.keys_out);
    const unsigned* vals = (const unsigned*)(ws + lo.sort.vals_out);
    const unsigned* bag_of = (const unsigned*)(ws + lo.sort.bag_of);
    float* ef = (float*)(ws + lo.edge_first);
    float* el = (float*)(ws + lo.edge_last);
    Shape s;
    rc = pick(D, vec_ok, &s);
    if (rc) return rc;
    // gradient rows in flight per lane group (tuning builds: env DLRM_ADAGRAD_KC = 1 | 2 | 4 | 8, default 2: measured 3.62 / 3.68 / 3.98 ms for 2 / 4 / 8 on the MLPerf-v2 batch): with the group-level resolve the
                               // rows depend on registers only, so more in flight is more bandwidth (before it, 1 was best: see the kernel's comment)
    static const int kc = DLRM_TUNE_ENV("DLRM_ADAGRAD_KC", 2);
    const size_t groups = (L + kG - 1) / kG;
    const int gpb = 256 / s.lpb;
    dim3 grid((unsigned)((groups + gpb - 1) / gpb), 1, 1), block(256);
#define ADA(V, LP, NC)                                                                                                         \
    do {                                                                                                                       \
        if (kc == 1) hipLaunchKernelGGL((adagrad_groups_kernel<V, LP, NC, KT, 1>), grid, block, 0, st, sa, aa, (long long)L, D, row_bits, keys, \
                           vals, bag_of, dout, (long long)dout_ld, clr, eps, ef, el);                                          \
        else if (kc == 2) hipLaunchKernelGGL((adagrad_groups_kernel<V, LP, NC, KT, 2>), grid, block, 0, st, sa, aa, (long long)L, D, row_bits, keys, \
                           vals, bag_of, dout, (long long)dout_ld, clr, eps, ef, el);                                          \
        else if (kc == 8) hipLaunchKernelGGL((adagrad_groups_kernel<V, LP, NC, KT, (LP >= 8 ? 8 : 4)>), grid, block, 0, st, sa, aa, (long long)L, D, row_bits, keys, \
                           vals, bag_of, dout, (long long)dout_ld, clr, eps, ef, el);                                          \
        else hipLaunchKernelGGL((adagrad_groups_kernel<V, LP, NC, KT, 4>), grid, block, 0, st, sa, aa, (long long)L, D, row_bits, keys, \
                           vals, bag_of, dout, (long long)dout_ld, clr, eps, ef, el);                                          \
        DLRM_LAUNCH_CHECK();                                                                                                   \
        hipLaunchKernelGGL((adagrad_fixup_kernel<V, LP, NC, KT>), grid, block, 0, st, sa, aa, (long long)L, D, row_bits, keys,  \
                           clr, eps, (const float*)ef, (const float*)el);                                                      \
        DLRM_LAUNCH_CHECK();                                                                                                   \
    } while (0)
    const int key = s.vec * 10000 + s.lpb * 10 + s.nch;
    switch (key) {
        case 40041: ADA(4, 4, 1); break;   case 40081: ADA(4, 8, 1); break;   case 40161: ADA(4, 16, 1); break;
        case 40321: ADA(4, 32, 1); break;  case 40641: ADA(4, 64, 1); break;  case 40642: ADA(4, 64, 2); break;
        case 40644: ADA(4, 64, 4); break;
        case 10041: ADA(1, 4, 1); break;   case 10081: ADA(1, 8, 1); break;   case 10161: ADA(1, 16, 1); break;
        case 10321: ADA(1, 32, 1); break;  case 10641: ADA(1, 64, 1); break;  case 10642: ADA(1, 64, 2); break;
        case 10644: ADA(1, 64, 4); break;
        default: return DLRM_E_RANGE;
    }
#undef ADA
    return 0;
}

__global__ __launch_bounds__(256) void adagrad_dense_kernel(long long n, float* __restrict__ w, float* __restrict__ sum,
                                                            const float* __restrict__ g, DlrmStep clr_, float eps) {
    const float clr = clr_;              // (by value, or read from the device scalar: common.h DlrmStep)
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float s = __builtin_fmaf(gi, gi, sum[i]);          // addcmul_(grad, grad, value=1)
        sum[i] = s;
        w[i] = __builtin_fmaf(-clr, gi / (sqrtf(s) + eps), w[i]); // addcdiv_(grad, std, value=-clr)
    }
}

}  // namespace

extern "C" int64_t dlrm_emb_adagrad_workspace_bytes(int T, int D, const int64_t* nnz_host, const int64_t* rows_host) {
    if (T <= 0 || D <= 0 || !nnz_host || !rows_host) return 0;
    size_t worst = 0;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        size_t L = 0; long long max_rows = 1;
        for (int k = 0; k < n; ++k) { L += (size_t)nnz_host[t0 + k]; if (rows_host[t0 + k] > max_rows) max_rows = rows_host[t0 + k]; }
        if (L == 0) continue;
        const int row_bits = bits_for(max_rows), key_bits = row_bits + bits_for(n);
        AdaLayout lo;
        if (ada_layout(L, key_bits > 32, key_bits, D, &lo, n, nnz_host + t0, rows_host + t0) != 0) return -1;
        if (lo.total > worst) worst = lo.total;
    }
    return (int64_t)worst;
}

extern "C" int dlrm_emb_bwd_rowwise_adagrad(int T, int64_t B, int D, void* const* weight_host,
                                            void* const* state_host, const int64_t* rows_host,
                                            const void* const* indices_host, const void* const* offsets_host,
                                            const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                                            const float* dout, int64_t dout_ld, float lr, const float* lr_dev, float eps,
                                            void* workspace, int64_t workspace_bytes, int64_t* err, void* stream) {
    if (T <= 0 || B <= 0 || D <= 0 || !weight_host || !state_host || !rows_host || !indices_host || !offsets_host || !nnz_host ||
        !dout || dout_ld < (int64_t)T * D)
        return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    bool vec_ok = dlrm_aligned16(dout) && (dout_ld % 4 == 0);
    for (int t = 0; t < T; ++t) {
        if (!weight_host[t] || !state_host[t]) return DLRM_E_ARG;
        vec_ok = vec_ok && dlrm_aligned16(weight_host[t]);
    }
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        int ids[DLRM_MAX_TABLES_PER_LAUNCH];
        size_t L = 0; long long max_rows = 1;
        for (int k = 0; k < n; ++k) {
            ids[k] = t0 + k; L += (size_t)nnz_host[t0 + k];
            if (rows_host[t0 + k] > max_rows) max_rows = rows_host[t0 + k];
        }
        if (L == 0) continue;
        if (L >= ((size_t)1 << 32)) return DLRM_E_RANGE;
        const int row_bits = bits_for(max_rows), key_bits = row_bits + bits_for(n);
        const bool wide = key_bits > 32;
        AdaLayout lo;
        int rc = ada_layout(L, wide, key_bits, D, &lo, n, nnz_host + t0, rows_host + t0);
        if (rc) return rc;
        if (!workspace || !dlrm_aligned16(workspace) || (size_t)workspace_bytes < lo.total) {
            fprintf(stderr, "libdlrm_hip: dlrm_emb_bwd_rowwise_adagrad: workspace too small (%lld < %zu bytes)\n",
                    (long long)workspace_bytes, lo.total);
            return DLRM_E_ARG;
        }
        rc = wide ? run_adagrad<unsigned long long>(n, ids, B, D, weight_host, state_host, rows_host, indices_host, offsets_host,
                                                    nnz_host, psw_host, idx_bits, dout, dout_ld, dlrm_step_pos(lr, lr_dev), eps, (char*)workspace, lo, L,
                                                    row_bits, key_bits, vec_ok, st, err)
                  : run_adagrad<unsigned>(n, ids, B, D, weight_host, state_host, rows_host, indices_host, offsets_host, nnz_host,
                                          psw_host, idx_bits, dout, dout_ld, dlrm_step_pos(lr, lr_dev), eps, (char*)workspace, lo, L, row_bits, key_bits,
                                          vec_ok, st, err);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int dlrm_adagrad_dense(int64_t n, float* w, float* sum, const float* g, float lr, const float* lr_dev, float eps, void* stream) {
    if (n <= 0 || !w || !sum || !g) return DLRM_E_ARG;
    long long nblk = (n + 255) / 256; if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(adagrad_dense_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (long long)n, w, sum, g, dlrm_step_pos(lr, lr_dev), eps);
    DLRM_LAUNCH_CHECK();
    return 0;
}
