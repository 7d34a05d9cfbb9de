// emb.hip — batched multi-table EmbeddingBag(sum) forward and fused backward+SGD for gfx950.
//
// Reference call sites replaced (see include/dlrm_hip.h):
//   forward : DLRM_Net.apply_emb loop over nn.EmbeddingBag            dlrm_s_pytorch.py:407-462
//   backward: EmbeddingBagBackward -> sparse COO grad -> SGD.step     dlrm_s_pytorch.py:1613,1620
//
// Design (HBM-bound integer/byte work, no MFMA):
//   * one launch covers every table: blockIdx.y = table, table pointers live in the kernarg
//     segment (EmbArgs by value), so there is no pointer-table H2D copy and the launch is
//     hipGraph-capturable.
//   * a "group" of LPB lanes owns one bag; lane c of the group owns columns [4c, 4c+4) of the
//     row (one 16-byte global_load_dwordx4 per row), so a D=128 row is one fully coalesced
//     512-byte read by half a wavefront and rows are accumulated IN INDEX ORDER per column —
//     bit-identical to the torch CPU kernel (no cross-lane reduction over rows).
//   * each group works on U bags at once: the U first-row loads are issued back to back
//     (U independent 512 B reads in flight per half-wave) before any dependent add, which is
//     what keeps HBM busy for one-hot (Criteo) inputs where every bag has exactly one row.
//   * bags with more rows continue with a 4-deep load pipeline per bag.
#include <stdlib.h>
#include "common.h"

namespace {

template <int VEC> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };

__device__ __forceinline__ void v_zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void v_zero(float& a) { a = 0.f; }
// acc = fma(w, v, acc) per component.  With w == 1.0f this is exactly acc + v (single rounding),
// so the unweighted path reproduces the reference's plain in-order sum bit for bit.
__device__ __forceinline__ void v_fma(float4& a, float w, const float4& v) {
    a.x = __builtin_fmaf(w, v.x, a.x); a.y = __builtin_fmaf(w, v.y, a.y);
    a.z = __builtin_fmaf(w, v.z, a.z); a.w = __builtin_fmaf(w, v.w, a.w);
}
__device__ __forceinline__ void v_fma(float& a, float w, const float& v) { a = __builtin_fmaf(w, v, a); }
__device__ __forceinline__ float4 v_scale(float s, const float4& v) {
    return make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
}
__device__ __forceinline__ float v_scale(float s, const float& v) { return s * v; }

__device__ __forceinline__ void v_atomic_add(float* p, const float4& v) {
    // -munsafe-fp-atomics: each of these is one global_atomic_add_f32 (no return, no CAS loop)
    atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
__device__ __forceinline__ void v_atomic_add(float* p, const float& v) { atomicAdd(p, v); }

// -------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------
// LOOP (dlrm_emb_fwd_pred only): a capped grid walks the bags with a grid stride.  A predicated launch that does NOT run still has its
// workgroups dispatched, and the plain launch has one workgroup per 16 bags (106 k workgroups at Criteo-Terabyte shapes: ~20 us of dispatch
// for a launch that returns at once); LOOP = false is the straight-line kernel of every other call, unchanged.
template <int VEC, int LPB, int NCH, typename IT, int U, bool LOOP = false>
__global__ __launch_bounds__(256) void emb_fwd_kernel(EmbArgs a, long long B, int D,
                                                      float* __restrict__ out, long long out_ld) {
    using VT = typename Vec<VEC>::T;
    if (a.pred.skip()) return;      // (dlrm_emb_fwd_pred: the other implementation of this step runs instead)
    const int t = blockIdx.y;
    const float* __restrict__ W = a.w[t];
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const float* __restrict__ psw = a.psw[t];
    const long long nnz = a.nnz[t];
    const long long rows = a.rows[t];

    constexpr int GPB = 256 / LPB;  // groups (bags in flight) per workgroup
    const int g = threadIdx.x / LPB;
    const int lig = threadIdx.x % LPB;
    long long b0 = ((long long)blockIdx.x * GPB + g) * U;
    if (b0 >= B) return;
  for (;;) {
    long long s[U], e[U];
    {
        long long o[U + 1];
#pragma unroll
        for (int u = 0; u <= U; ++u) {
            const long long b = b0 + u;
            o[u] = (b < B) ? (long long)off[b] : nnz;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s[u] = o[u]; e[u] = (b0 + u < B) ? o[u + 1] : o[u]; }
    }

    VT acc[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) v_zero(acc[u][c]);

    // ---- phase 1: first row of every bag, all loads in flight together --------------------
    // an out-of-range index contributes nothing and is reported (the reference raises on it)
    long long r0[U];
    float w0[U];
    bool ok0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        r0[u] = 0; w0[u] = 1.f; ok0[u] = false;
        if (s[u] < e[u]) {
            r0[u] = (long long)idx[s[u]];
            if (psw) w0[u] = psw[s[u]];
            ok0[u] = dlrm_index_ok(r0[u], rows);
            if (!ok0[u]) dlrm_report_bad_index(a.err, a.slot[t], r0[u], rows);
        }
    }
    VT v0[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (c * LPB + lig) * VEC;
            v_zero(v0[u][c]);
            if (ok0[u] && col < D) v0[u][c] = *(const VT*)(W + r0[u] * D + col);
        }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (ok0[u]) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) v_fma(acc[u][c], w0[u], v0[u][c]);
        }

    // ---- phase 2: remaining rows of multi-hot bags, 4 row loads in flight per bag ----------
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long long i = s[u] + 1;
        const long long end = e[u];
        for (; i + 4 <= end; i += 4) {
            long long r[4]; float w[4]; VT v[4][NCH];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = (long long)idx[i + k]; w[k] = psw ? psw[i + k] : 1.f;
                ok[k] = dlrm_index_ok(r[k], rows);
                if (!ok[k]) dlrm_report_bad_index(a.err, a.slot[t], r[k], rows);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int col = (c * LPB + lig) * VEC;
                    v_zero(v[k][c]);
                    if (ok[k] && col < D) v[k][c] = *(const VT*)(W + r[k] * D + col);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ok[k]) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) v_fma(acc[u][c], w[k], v[k][c]);
                }
            }
        }
        for (; i < end; ++i) {
            const long long r = (long long)idx[i];
            const float w = psw ? psw[i] : 1.f;
            if (!dlrm_index_ok(r, rows)) { dlrm_report_bad_index(a.err, a.slot[t], r, rows); continue; }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (c * LPB + lig) * VEC;
                if (col < D) { VT v = *(const VT*)(W + r * D + col); v_fma(acc[u][c], w, v); }
            }
        }
    }

    // ---- store: bag b of table t goes to out[b, (slot_base+t)*D : +D] ------------------------
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long b = b0 + u;
        if (b < B) {
            float* o = out + b * out_ld + (long long)a.slot[t] * D;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (c * LPB + lig) * VEC;
                if (col < D) *(VT*)(o + col) = acc[u][c];
            }
        }
    }
    if constexpr (!LOOP) break;
    b0 += (long long)gridDim.x * GPB * U;
    if (b0 >= B) break;
  }
}

// -------------------------------------------------------------------------------------------
// backward + SGD, atomic mode, general tables:  W[idx,:] += (-lr * psw) * dout[b,:]
// -------------------------------------------------------------------------------------------
template <int VEC, int LPB, int NCH, typename IT, int U>
__global__ __launch_bounds__(256) void emb_bwd_sgd_atomic_kernel(EmbArgs a, long long B, int D,
                                                                 const float* __restrict__ dout,
                                                                 long long dout_ld, DlrmStep neg_lr_) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    using VT = typename Vec<VEC>::T;
    const int t = blockIdx.y;
    float* __restrict__ W = a.w[t];
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const float* __restrict__ psw = a.psw[t];
    const long long nnz = a.nnz[t];

    constexpr int GPB = 256 / LPB;
    const int g = threadIdx.x / LPB;
    const int lig = threadIdx.x % LPB;
    const long long b0 = ((long long)blockIdx.x * GPB + g) * U;
    if (b0 >= B) return;

    long long s[U], e[U];
    {
        long long o[U + 1];
#pragma unroll
        for (int u = 0; u <= U; ++u) {
            const long long b = b0 + u;
            o[u] = (b < B) ? (long long)off[b] : nnz;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s[u] = o[u]; e[u] = (b0 + u < B) ? o[u + 1] : o[u]; }
    }
    // gradient rows (coalesced 16 B per lane) and first indices, all in flight together
    VT gr[U][NCH];
    long long r0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long b = b0 + u;
        r0[u] = (s[u] < e[u]) ? (long long)idx[s[u]] : 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = (c * LPB + lig) * VEC;
            v_zero(gr[u][c]);
            if (b < B && col < D)
                gr[u][c] = v_scale(neg_lr, *(const VT*)(dout + b * dout_ld + (long long)a.slot[t] * D + col));
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        for (long long i = s[u]; i < e[u]; ++i) {
            const long long r = (i == s[u]) ? r0[u] : (long long)idx[i];
            const float w = psw ? psw[i] : 1.f;
            if (!dlrm_index_ok(r, a.rows[t])) { dlrm_report_bad_index(a.err, a.slot[t], r, a.rows[t]); continue; }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = (c * LPB + lig) * VEC;
                if (col < D) v_atomic_add(W + r * D + col, psw ? v_scale(w, gr[u][c]) : gr[u][c]);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// backward + SGD, atomic mode, TINY tables (rows*D*4 <= LDS budget): every workgroup first sums
// its slice of the batch into an LDS image of the whole table gradient (LDS atomics), then
// flushes the non-zero entries with one global atomic each.  Criteo has tables with 3..100 rows
// that are hit ~B/rows times per step; without this the same few cache lines take tens of
// thousands of serialized global atomics.
// -------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void emb_bwd_sgd_lds_kernel(EmbArgs a, long long B, int D,
                                                              const float* __restrict__ dout,
                                                              long long dout_ld, DlrmStep neg_lr_,
                                                              int bags_per_block) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    extern __shared__ __attribute__((aligned(16))) float lds_acc[];
    const int t = blockIdx.y;
    float* __restrict__ W = a.w[t];
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const float* __restrict__ psw = a.psw[t];
    const long long nnz = a.nnz[t];
    const int n_elem = (int)(a.rows[t] * D);

    for (int e = threadIdx.x; e < n_elem; e += 256) lds_acc[e] = 0.f;
    __syncthreads();

    const long long b_begin = (long long)blockIdx.x * bags_per_block;
    const long long b_end = (b_begin + bags_per_block < B) ? b_begin + bags_per_block : B;
    // thread -> (bag, column): consecutive threads take consecutive columns of one bag so the
    // dout reads are coalesced and the LDS atomics of a wave land on distinct banks.
    const long long n_work = (b_end - b_begin) * D;
    for (long long x = threadIdx.x; x < n_work; x += 256) {
        const long long bl = x / D;
        const int d = (int)(x - bl * D);
        const long long b = b_begin + bl;
        const float gneg = neg_lr * dout[b * dout_ld + (long long)a.slot[t] * D + d];
        const long long s = (long long)off[b];
        const long long e = (b + 1 < B) ? (long long)off[b + 1] : nnz;
        for (long long i = s; i < e; ++i) {
            const long long r = (long long)idx[i];
            if (!dlrm_index_ok(r, a.rows[t])) { dlrm_report_bad_index(a.err, a.slot[t], r, a.rows[t]); continue; }
            atomicAdd(&lds_acc[r * D + d], psw ? psw[i] * gneg : gneg);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_elem; e += 256) {
        const float v = lds_acc[e];
        if (v != 0.f) atomicAdd(W + e, v);
    }
}

// -------------------------------------------------------------------------------------------
// backward + SGD, deterministic mode: owner-computes.  Group G of NG owns the rows with
// row % NG == G, scans every lookup of its table in input order and applies
//     W[r,:] = fma(-lr, psw_i * dout[bag(i),:], W[r,:])
// sequentially — the exact operation order of `p.add_(sparse_grad, alpha=-lr)` on the
// uncoalesced COO gradient, hence bit-identical to the reference.  O(NG * nnz) index reads:
// a validation mode, not the production path.
// -------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void emb_bwd_sgd_det_kernel(EmbArgs a, long long B, int D,
                                                              const float* __restrict__ dout,
                                                              long long dout_ld, DlrmStep neg_lr_) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    const int t = blockIdx.y;
    float* __restrict__ W = a.w[t];
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    const float* __restrict__ psw = a.psw[t];
    const long long nnz = a.nnz[t];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long NG = (long long)gridDim.x * 4;
    const long long G = (long long)blockIdx.x * 4 + wave;  // one owner group = one wavefront
    for (long long b = 0; b < B; ++b) {
        const long long s = (long long)off[b];
        const long long e = (b + 1 < B) ? (long long)off[b + 1] : nnz;
        for (long long i = s; i < e; ++i) {
            const long long r = (long long)idx[i];
            if (!dlrm_index_ok(r, a.rows[t])) { dlrm_report_bad_index(a.err, a.slot[t], r, a.rows[t]); continue; }
            if (r % NG != G) continue;  // wave-uniform
            const float w = psw ? psw[i] : 1.f;
            for (int d = lane; d < D; d += 64) {
                float gval = dout[b * dout_ld + (long long)a.slot[t] * D + d];
                if (psw) gval = gval * w;  // the COO value the reference materialises
                float* p = W + r * D + d;
                *p = __builtin_fmaf(neg_lr, gval, *p);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// backward WITHOUT the fused update: the reference's sparse COO gradient, materialised.
//   values_t[i, :] = psw_t[i] * dout[bag(i), t*D:(t+1)*D]      for every lookup i of table t, in input order
// (indices of the COO tensor are the lookup indices verbatim, uncoalesced — what EmbeddingBagBackward returns,
// dlrm_s_pytorch.py:1613).  Escape hatch for optimizers the fused kernels do not implement: 2R extra bytes per lookup.
// -------------------------------------------------------------------------------------------
struct CooArgs { float* values[DLRM_MAX_TABLES_PER_LAUNCH]; };

template <typename IT>
__global__ __launch_bounds__(256) void emb_bwd_coo_kernel(EmbArgs a, CooArgs ca, long long B, int D,
                                                          const float* __restrict__ dout, long long dout_ld) {
    const int t = blockIdx.y;
    const IT* __restrict__ off = (const IT*)a.off[t];
    const float* __restrict__ psw = a.psw[t];
    float* __restrict__ values = ca.values[t];
    const long long nnz = a.nnz[t];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + wave;          // one wavefront per bag
    if (b >= B) return;
    const long long s = (long long)off[b];
    const long long e = (b + 1 < B) ? (long long)off[b + 1] : nnz;
    const float* g = dout + b * dout_ld + (long long)a.slot[t] * D;
    for (int d = lane; d < D; d += 64) {
        const float gv = g[d];
        for (long long i = s; i < e; ++i) values[i * D + d] = psw ? gv * psw[i] : gv;
    }
}

// -------------------------------------------------------------------------------------------
// learned / fixed pooling weights (--weighted-pooling, dlrm_s_pytorch.py:289-293, 425-428):
//   forward   psw_t[i] = vW_t[idx_t[i]]                                  (`v_W_l[k].gather(0, indices)`)
//   backward  dvW_t[idx_t[i]] += < dout[bag(i), t*D:(t+1)*D], W_t[idx_t[i], :] >      (EmbeddingBag's per_sample_weights
//             gradient followed by the gather's scatter-add, fused: one wavefront per bag, a wave-wide dot per lookup)
// -------------------------------------------------------------------------------------------
struct PoolArgs { float* vw[DLRM_MAX_TABLES_PER_LAUNCH]; float* out[DLRM_MAX_TABLES_PER_LAUNCH]; };

template <typename IT>
__global__ __launch_bounds__(256) void pool_weights_gather_kernel(EmbArgs a, PoolArgs pa) {
    const int t = blockIdx.y;
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const float* __restrict__ vw = pa.vw[t];
    float* __restrict__ out = pa.out[t];
    const long long n = a.nnz[t], rows = a.rows[t];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = (long long)idx[i];
        float v = 0.f;
        if (dlrm_index_ok(r, rows)) v = vw[r]; else dlrm_report_bad_index(a.err, a.slot[t], r, rows);
        out[i] = v;
    }
}

template <typename IT>
__global__ __launch_bounds__(256) void emb_psw_grad_kernel(EmbArgs a, PoolArgs pa, long long B, int D,
                                                           const float* __restrict__ dout, long long dout_ld) {
    const int t = blockIdx.y;
    const float* __restrict__ W = a.w[t];
    const IT* __restrict__ idx = (const IT*)a.idx[t];
    const IT* __restrict__ off = (const IT*)a.off[t];
    float* __restrict__ dvw = pa.vw[t];
    const long long nnz = a.nnz[t], rows = a.rows[t];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + wave;
    if (b >= B) return;
    const long long s = (long long)off[b];
    const long long e = (b + 1 < B) ? (long long)off[b + 1] : nnz;
    const float* g = dout + b * dout_ld + (long long)a.slot[t] * D;
    for (long long i = s; i < e; ++i) {
        const long long r = (long long)idx[i];
        if (!dlrm_index_ok(r, rows)) continue;                 // already reported by the forward pass
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc = __builtin_fmaf(g[d], W[r * D + d], acc);
        acc = dlrm_wave_sum(acc);
        if (lane == 0) atomicAdd(dvw + r, acc);
    }
}

// -------------------------------------------------------------------------------------------
// host-side dispatch
// -------------------------------------------------------------------------------------------
struct Shape { int vec, lpb, nch; };

static int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

static bool pick_shape(int D, bool vec_ok, Shape* s) {
    if (vec_ok && D % 4 == 0) {
        const int d4 = D / 4;
        int lpb = pow2ceil(d4); if (lpb < 4) lpb = 4; if (lpb > 64) lpb = 64;
        const int nch = (d4 + lpb - 1) / lpb;
        if (nch > 4) return false;
        *s = {4, lpb, nch == 3 ? 4 : nch};
        return true;
    }
    int lpb = pow2ceil(D); if (lpb < 4) lpb = 4; if (lpb > 64) lpb = 64;
    const int nch = (D + lpb - 1) / lpb;
    if (nch > 4) return false;
    *s = {1, lpb, nch == 3 ? 4 : nch};
    return true;
}

constexpr int kU = 4;  // bags per group

#define EMB_DISPATCH_SHAPE(KERNEL, IT, ...)                                                        \
    do {                                                                                           \
        const int key = sh.vec * 10000 + sh.lpb * 10 + sh.nch;                                     \
        switch (key) {                                                                             \
            case 4 * 10000 + 4 * 10 + 1:  hipLaunchKernelGGL((KERNEL<4, 4, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break;  \
            case 4 * 10000 + 8 * 10 + 1:  hipLaunchKernelGGL((KERNEL<4, 8, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break;  \
            case 4 * 10000 + 16 * 10 + 1: hipLaunchKernelGGL((KERNEL<4, 16, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 4 * 10000 + 32 * 10 + 1: hipLaunchKernelGGL((KERNEL<4, 32, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 4 * 10000 + 64 * 10 + 1: hipLaunchKernelGGL((KERNEL<4, 64, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 4 * 10000 + 64 * 10 + 2: hipLaunchKernelGGL((KERNEL<4, 64, 2, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 4 * 10000 + 64 * 10 + 4: hipLaunchKernelGGL((KERNEL<4, 64, 4, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 1 * 10000 + 4 * 10 + 1:  hipLaunchKernelGGL((KERNEL<1, 4, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break;  \
            case 1 * 10000 + 8 * 10 + 1:  hipLaunchKernelGGL((KERNEL<1, 8, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break;  \
            case 1 * 10000 + 16 * 10 + 1: hipLaunchKernelGGL((KERNEL<1, 16, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 1 * 10000 + 32 * 10 + 1: hipLaunchKernelGGL((KERNEL<1, 32, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 1 * 10000 + 64 * 10 + 1: hipLaunchKernelGGL((KERNEL<1, 64, 1, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 1 * 10000 + 64 * 10 + 2: hipLaunchKernelGGL((KERNEL<1, 64, 2, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            case 1 * 10000 + 64 * 10 + 4: hipLaunchKernelGGL((KERNEL<1, 64, 4, IT, kU>), grid, block, 0, st, __VA_ARGS__); break; \
            default: return DLRM_E_RANGE;                                                          \
        }                                                                                          \
    } while (0)

static int check_common(int T, int64_t B, int D, const void* const* weight_host, const int64_t* rows_host,
                        const void* const* indices_host, const void* const* offsets_host,
                        const int64_t* nnz_host, int idx_bits) {
    if (T <= 0 || B <= 0 || D <= 0) return DLRM_E_ARG;
    if (!weight_host || !rows_host || !indices_host || !offsets_host || !nnz_host) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    for (int t = 0; t < T; ++t) {
        if (!weight_host[t] || !offsets_host[t]) return DLRM_E_ARG;
        if (nnz_host[t] < 0 || rows_host[t] <= 0) return DLRM_E_ARG;
        if (nnz_host[t] > 0 && !indices_host[t]) return DLRM_E_ARG;
    }
    return 0;
}

static void fill_args(EmbArgs& a, const int* ids, int n, void* const* weight_host, const int64_t* rows_host,
                      const void* const* indices_host, const void* const* offsets_host,
                      const int64_t* nnz_host, const void* const* psw_host, int64_t* err) {
    a.err = (long long*)err;
    a.pred.flag = nullptr; a.pred.nonzero = 0;
    for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) {
        const int t = ids[k < n ? k : 0];
        a.w[k] = (float*)weight_host[t];
        a.idx[k] = indices_host[t];
        a.off[k] = offsets_host[t];
        a.psw[k] = psw_host ? (const float*)psw_host[t] : nullptr;
        a.nnz[k] = nnz_host[t];
        a.rows[k] = rows_host[t];
        a.slot[k] = t;
    }
}

}  // namespace

static int emb_fwd_impl(int T, int64_t B, int D, const void* const* weight_host,
                        const int64_t* rows_host, const void* const* indices_host,
                        const void* const* offsets_host, const int64_t* nnz_host,
                        const void* const* psw_host, int idx_bits, float* out, int64_t out_ld,
                        int64_t* err, DlrmPred pred, void* stream);

extern "C" int dlrm_emb_fwd(int T, int64_t B, int D, const void* const* weight_host,
                            const int64_t* rows_host, const void* const* indices_host,
                            const void* const* offsets_host, const int64_t* nnz_host,
                            const void* const* psw_host, int idx_bits, float* out, int64_t out_ld,
                            int64_t* err, void* stream) {
    return emb_fwd_impl(T, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, idx_bits, out, out_ld, err,
                        DlrmPred{nullptr, 0}, stream);
}

extern "C" int dlrm_emb_fwd_pred(int T, int64_t B, int D, const void* const* weight_host,
                                 const int64_t* rows_host, const void* const* indices_host,
                                 const void* const* offsets_host, const int64_t* nnz_host,
                                 const void* const* psw_host, int idx_bits, float* out, int64_t out_ld,
                                 int64_t* err, const int32_t* pred_flag, int pred_nonzero, void* stream) {
    return emb_fwd_impl(T, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, idx_bits, out, out_ld, err,
                        DlrmPred{(const int*)pred_flag, pred_nonzero}, stream);
}

static int emb_fwd_impl(int T, int64_t B, int D, const void* const* weight_host,
                        const int64_t* rows_host, const void* const* indices_host,
                        const void* const* offsets_host, const int64_t* nnz_host,
                        const void* const* psw_host, int idx_bits, float* out, int64_t out_ld,
                        int64_t* err, DlrmPred pred, void* stream) {
    int rc = check_common(T, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host, idx_bits);
    if (rc) return rc;
    if (!out || out_ld < (int64_t)T * D) return DLRM_E_ARG;
    hipStream_t st = (hipStream_t)stream;

    bool vec_ok = dlrm_aligned16(out) && (out_ld % 4 == 0);
    for (int t = 0; t < T; ++t) vec_ok = vec_ok && dlrm_aligned16(weight_host[t]);
    Shape sh;
    if (!pick_shape(D, vec_ok, &sh)) {
        fprintf(stderr, "libdlrm_hip: dlrm_emb_fwd: embedding dim %d not supported (max 1024, or 256 unaligned)\n", D);
        return DLRM_E_RANGE;
    }
    const int bags_per_block = (256 / sh.lpb) * kU;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        int ids[DLRM_MAX_TABLES_PER_LAUNCH];
        for (int k = 0; k < n; ++k) ids[k] = t0 + k;
        EmbArgs a;
        fill_args(a, ids, n, (void* const*)weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, err);
        a.pred = pred;
        dim3 grid((unsigned)((B + bags_per_block - 1) / bags_per_block), (unsigned)n, 1), block(256, 1, 1);
        if (pred.flag) {           // predicated: the D = 128 shape only (what the fused path's fallback needs), capped grid + grid stride
            if (!(sh.vec == 4 && sh.lpb == 32 && sh.nch == 1)) return DLRM_E_MODE;
            dim3 gl(grid.x < 512u ? grid.x : 512u, (unsigned)n, 1);
            if (idx_bits == 64) hipLaunchKernelGGL((emb_fwd_kernel<4, 32, 1, long long, 2, true>), gl, block, 0, st, a, (long long)B, D, out, (long long)out_ld);
            else                hipLaunchKernelGGL((emb_fwd_kernel<4, 32, 1, int, 2, true>), gl, block, 0, st, a, (long long)B, D, out, (long long)out_ld);
            DLRM_LAUNCH_CHECK();
            continue;
        }
        // bags per lane group for the D = 128 shape: 2 (default; env DLRM_EMB_FWD_U = 1 | 2 | 4 | 8).  Measured on one box at
        // Criteo-Terabyte shapes: U = 1 0.356 ms, 2 0.282, 4 0.315, 8 0.312 — two independent row loads per half-wave and twice
        // the waves beat four loads per half-wave
        static const int u_alt = DLRM_TUNE_ENV("DLRM_EMB_FWD_U", 2);      // (the variable exists in tuning builds only)
        if (u_alt && sh.vec == 4 && sh.lpb == 32 && sh.nch == 1 && (u_alt == 1 || u_alt == 2 || u_alt == 8)) {
            const int bpb = (256 / 32) * u_alt;
            dim3 g2((unsigned)((B + bpb - 1) / bpb), (unsigned)n, 1);
#define EMB_FWD_U(UU) do { if (idx_bits == 64) hipLaunchKernelGGL((emb_fwd_kernel<4, 32, 1, long long, UU>), g2, block, 0, st, a, (long long)B, D, out, (long long)out_ld); \
                           else hipLaunchKernelGGL((emb_fwd_kernel<4, 32, 1, int, UU>), g2, block, 0, st, a, (long long)B, D, out, (long long)out_ld); } while (0)
            if (u_alt == 1) EMB_FWD_U(1); else if (u_alt == 2) EMB_FWD_U(2); else EMB_FWD_U(8);
#undef EMB_FWD_U
            DLRM_LAUNCH_CHECK();
            continue;
        }
        if (idx_bits == 64) EMB_DISPATCH_SHAPE(emb_fwd_kernel, long long, a, (long long)B, D, out, (long long)out_ld);
        else                EMB_DISPATCH_SHAPE(emb_fwd_kernel, int, a, (long long)B, D, out, (long long)out_ld);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dlrm_emb_bwd_sgd(int T, int64_t B, int D, void* const* weight_host,
                                const int64_t* rows_host, const void* const* indices_host,
                                const void* const* offsets_host, const int64_t* nnz_host,
                                const void* const* psw_host, int idx_bits, const float* dout,
                                int64_t dout_ld, float lr, const float* lr_dev, int mode, void* workspace,
                                int64_t workspace_bytes, int64_t* err, void* stream) {
    int rc = check_common(T, B, D, (const void* const*)weight_host, rows_host, indices_host, offsets_host,
                          nnz_host, idx_bits);
    if (rc) return rc;
    if (!dout || dout_ld < (int64_t)T * D) return DLRM_E_ARG;
    if (mode != DLRM_UPD_ATOMIC && mode != DLRM_UPD_DETERMINISTIC && mode != DLRM_UPD_SORTED) return DLRM_E_MODE;
    if (mode == DLRM_UPD_SORTED)
        return dlrm_emb_bwd_sgd_sorted_impl(T, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host,
                                            psw_host, idx_bits, dout, dout_ld, lr, lr_dev, workspace, workspace_bytes, err, stream);
    hipStream_t st = (hipStream_t)stream;
    const DlrmStep neg_lr = dlrm_step_neg(lr, lr_dev);
    dim3 block(256, 1, 1);

    if (mode == DLRM_UPD_DETERMINISTIC) {
        for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
            const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
            int ids[DLRM_MAX_TABLES_PER_LAUNCH];
            for (int k = 0; k < n; ++k) ids[k] = t0 + k;
            EmbArgs a;
            fill_args(a, ids, n, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, err);
            dim3 grid(64, (unsigned)n, 1);
            if (idx_bits == 64)
                hipLaunchKernelGGL(emb_bwd_sgd_det_kernel<long long>, grid, block, 0, st, a, (long long)B, D, dout, (long long)dout_ld, neg_lr);
            else
                hipLaunchKernelGGL(emb_bwd_sgd_det_kernel<int>, grid, block, 0, st, a, (long long)B, D, dout, (long long)dout_ld, neg_lr);
            DLRM_LAUNCH_CHECK();
        }
        return 0;
    }

    // atomic mode: tables whose whole gradient image fits the LDS budget ("tiny") go through the
    // LDS pre-reduction kernel, everything else through direct global atomics; each class is
    // launched once per 32 tables (EmbArgs.slot keeps the original table -> dout column mapping).
    constexpr int64_t kLdsBudget = 64 * 1024;  // bytes of LDS gradient image per workgroup
    bool vec_ok = dlrm_aligned16(dout) && (dout_ld % 4 == 0);
    for (int t = 0; t < T; ++t) vec_ok = vec_ok && dlrm_aligned16(weight_host[t]);
    Shape sh;
    if (!pick_shape(D, vec_ok, &sh)) return DLRM_E_RANGE;
    const int bags_per_block = (256 / sh.lpb) * kU;

    for (int cls = 0; cls < 2; ++cls) {       // 0 = general, 1 = tiny
        int t = 0;
        while (t < T) {
            int ids[DLRM_MAX_TABLES_PER_LAUNCH];
            int n = 0;
            int64_t max_rows = 0;
            for (; t < T && n < DLRM_MAX_TABLES_PER_LAUNCH; ++t) {
                const bool tiny = rows_host[t] * (int64_t)D * 4 <= kLdsBudget;
                if ((int)tiny != cls || nnz_host[t] == 0) continue;
                ids[n++] = t;
                if (rows_host[t] > max_rows) max_rows = rows_host[t];
            }
            if (n == 0) break;
            EmbArgs a;
            fill_args(a, ids, n, weight_host, rows_host, indices_host, offsets_host, nnz_host, psw_host, err);
            if (cls == 1) {
                const size_t lds = (size_t)(max_rows * D * 4);
                // enough bags per workgroup to amortise the flush, enough workgroups to fill the chip
                int bpb = 1024;
                while (bpb > 64 && (B + bpb - 1) / bpb * n < 512) bpb >>= 1;
                dim3 grid((unsigned)((B + bpb - 1) / bpb), (unsigned)n, 1);
                if (idx_bits == 64)
                    hipLaunchKernelGGL(emb_bwd_sgd_lds_kernel<long long>, grid, block, lds, st, a, (long long)B, D, dout, (long long)dout_ld, neg_lr, bpb);
                else
                    hipLaunchKernelGGL(emb_bwd_sgd_lds_kernel<int>, grid, block, lds, st, a, (long long)B, D, dout, (long long)dout_ld, neg_lr, bpb);
            } else {
                dim3 grid((unsigned)((B + bags_per_block - 1) / bags_per_block), (unsigned)n, 1);
                if (idx_bits == 64) EMB_DISPATCH_SHAPE(emb_bwd_sgd_atomic_kernel, long long, a, (long long)B, D, dout, (long long)dout_ld, neg_lr);
                else                EMB_DISPATCH_SHAPE(emb_bwd_sgd_atomic_kernel, int, a, (long long)B, D, dout, (long long)dout_ld, neg_lr);
            }
            DLRM_LAUNCH_CHECK();
        }
    }
    return 0;
}

extern "C" int dlrm_emb_bwd_coo(int T, int64_t B, int D, const void* const* offsets_host, const int64_t* nnz_host,
                                const void* const* psw_host, int idx_bits, const float* dout, int64_t dout_ld,
                                void* const* values_host, void* stream) {
    if (T <= 0 || B <= 0 || D <= 0 || !offsets_host || !nnz_host || !values_host || !dout || dout_ld < (int64_t)T * D) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        EmbArgs a = {};
        CooArgs ca = {};
        for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) {
            const int t = t0 + (k < n ? k : 0);
            if (!offsets_host[t] || nnz_host[t] < 0 || (nnz_host[t] > 0 && !values_host[t])) return DLRM_E_ARG;
            a.off[k] = offsets_host[t]; a.psw[k] = psw_host ? (const float*)psw_host[t] : nullptr;
            a.nnz[k] = nnz_host[t]; a.slot[k] = t; ca.values[k] = (float*)values_host[t];
        }
        dim3 grid((unsigned)((B + 3) / 4), (unsigned)n, 1), block(256, 1, 1);
        if (idx_bits == 64) hipLaunchKernelGGL(emb_bwd_coo_kernel<long long>, grid, block, 0, st, a, ca, (long long)B, D, dout, (long long)dout_ld);
        else                hipLaunchKernelGGL(emb_bwd_coo_kernel<int>, grid, block, 0, st, a, ca, (long long)B, D, dout, (long long)dout_ld);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dlrm_pool_weights_gather(int T, const int64_t* rows_host, const void* const* indices_host, const int64_t* nnz_host,
                                        int idx_bits, const void* const* vw_host, void* const* psw_out_host, int64_t* err,
                                        void* stream) {
    if (T <= 0 || !rows_host || !indices_host || !nnz_host || !vw_host || !psw_out_host) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        EmbArgs a = {};
        PoolArgs pa = {};
        a.err = (long long*)err;
        long long most = 0;
        for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) {
            const int t = t0 + (k < n ? k : 0);
            if (nnz_host[t] < 0 || rows_host[t] <= 0 || (nnz_host[t] > 0 && (!indices_host[t] || !vw_host[t] || !psw_out_host[t]))) return DLRM_E_ARG;
            a.idx[k] = indices_host[t]; a.nnz[k] = k < n ? nnz_host[t] : 0; a.rows[k] = rows_host[t]; a.slot[k] = t;
            pa.vw[k] = (float*)vw_host[t]; pa.out[k] = (float*)psw_out_host[t];
            if (k < n && nnz_host[t] > most) most = nnz_host[t];
        }
        if (most == 0) continue;
        long long nb = (most + 255) / 256; if (nb > 1024) nb = 1024;
        dim3 grid((unsigned)nb, (unsigned)n, 1), block(256);
        if (idx_bits == 64) hipLaunchKernelGGL(pool_weights_gather_kernel<long long>, grid, block, 0, st, a, pa);
        else                hipLaunchKernelGGL(pool_weights_gather_kernel<int>, grid, block, 0, st, a, pa);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dlrm_emb_psw_grad(int T, int64_t B, int D, const void* const* weight_host, const int64_t* rows_host,
                                 const void* const* indices_host, const void* const* offsets_host, const int64_t* nnz_host,
                                 int idx_bits, const float* dout, int64_t dout_ld, void* const* dvw_host, void* stream) {
    int rc = check_common(T, B, D, weight_host, rows_host, indices_host, offsets_host, nnz_host, idx_bits);
    if (rc) return rc;
    if (!dout || dout_ld < (int64_t)T * D || !dvw_host) return DLRM_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int t = 0; t < T; ++t) {                    // dvW is an OUTPUT: zeroed here, then accumulated with atomics
        if (!dvw_host[t]) return DLRM_E_ARG;
        const int e = dlrm_zero2d((float*)dvw_host[t], rows_host[t], rows_host[t], 1, st);      // a kernel, not a memset node (common.h)
        if (e) return e;
    }
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        int ids[DLRM_MAX_TABLES_PER_LAUNCH];
        for (int k = 0; k < n; ++k) ids[k] = t0 + k;
        EmbArgs a;
        fill_args(a, ids, n, (void* const*)weight_host, rows_host, indices_host, offsets_host, nnz_host, nullptr, nullptr);
        PoolArgs pa = {};
        for (int k = 0; k < DLRM_MAX_TABLES_PER_LAUNCH; ++k) pa.vw[k] = (float*)dvw_host[t0 + (k < n ? k : 0)];
        dim3 grid((unsigned)((B + 3) / 4), (unsigned)n, 1), block(256);
        if (idx_bits == 64) hipLaunchKernelGGL(emb_psw_grad_kernel<long long>, grid, block, 0, st, a, pa, (long long)B, D, dout, (long long)dout_ld);
        else                hipLaunchKernelGGL(emb_psw_grad_kernel<int>, grid, block, 0, st, a, pa, (long long)B, D, dout, (long long)dout_ld);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

// -------------------------------------------------------------------------------------------
// "one lookup per bag" proof for the fused lookup + interaction path (dlrm_interact_*_gather): offsets_t[b] == b for every table
// and bag.  nnz == B alone does not prove it (an empty bag plus a two-lookup bag also sum to B, and EmbeddingBag accepts that:
// dlrm_s_pytorch.py:453-457).  One coalesced pass over T*B offsets; *violations (device-visible, e.g. pinned host memory) receives the
// number of bags whose start differs from their number — the caller synchronises the stream and reads it.
// -------------------------------------------------------------------------------------------
namespace {
struct IotaArgs { const void* off[DLRM_MAX_TABLES_PER_LAUNCH]; };
template <typename IdxT>
__global__ __launch_bounds__(256) void offsets_iota_kernel(IotaArgs a, long long B, int* __restrict__ violations, int* __restrict__ mirror) {
    const IdxT* off = (const IdxT*)a.off[blockIdx.y];
    int bad = 0;
    for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long long)gridDim.x * 256)
        bad += ((long long)off[b] != b);
    if (__any(bad != 0)) {
        atomicAdd(violations, bad);
        if (mirror && bad) *(volatile int*)mirror = 1;       // (a host-visible copy of "not zero": plain stores, any one of them wins)
    }
}
}  // namespace

static int offsets_iota_impl(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* violations, int32_t* mirror, void* stream);

extern "C" int dlrm_offsets_are_iota(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* violations, void* stream) {
    return offsets_iota_impl(T, B, offsets_host, idx_bits, violations, nullptr, stream);
}

extern "C" int dlrm_offsets_iota_flags(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* flag_dev, int32_t* flag_host,
                                       void* stream) {
    return offsets_iota_impl(T, B, offsets_host, idx_bits, flag_dev, flag_host, stream);
}

static int offsets_iota_impl(int T, int64_t B, const void* const* offsets_host, int idx_bits, int32_t* violations, int32_t* mirror, void* stream) {
    if (T <= 0 || B <= 0 || !offsets_host || !violations) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < T; t0 += DLRM_MAX_TABLES_PER_LAUNCH) {
        const int n = (T - t0 < DLRM_MAX_TABLES_PER_LAUNCH) ? T - t0 : DLRM_MAX_TABLES_PER_LAUNCH;
        IotaArgs a = {};
        for (int k = 0; k < n; ++k) { if (!offsets_host[t0 + k]) return DLRM_E_ARG; a.off[k] = offsets_host[t0 + k]; }
        long long nb = (B + 255) / 256; if (nb > 256) nb = 256;
        dim3 grid((unsigned)nb, (unsigned)n, 1), block(256);
        if (idx_bits == 64) hipLaunchKernelGGL(offsets_iota_kernel<long long>, grid, block, 0, st, a, (long long)B, (int*)violations, (int*)mirror);
        else                hipLaunchKernelGGL(offsets_iota_kernel<int>, grid, block, 0, st, a, (long long)B, (int*)violations, (int*)mirror);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}
