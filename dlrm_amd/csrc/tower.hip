// tower.hip — an MLP tower (nn.Sequential(Linear, act, Linear, act, ...), dlrm_s_pytorch.py:208-246, 399-405) for SMALL batches as
// four launches instead of one GEMM launch per layer and direction (plus a reduction launch per weight gradient):
//   tower_fwd_kernel     all layers forward: a workgroup owns 16 batch rows, the activations of those rows stay in LDS from layer to
//                        layer (each is also written out: the backward pass reads them), weights stream from L2 / HBM
//   tower_bwd_kernel     the data-gradient chain dZ_L -> dZ_{L-1} -> ... (-> dX), same ownership; every layer's dZ is written out
//   tower_wgrad_kernel   the weight / bias gradients of ALL layers in one launch: 64 x 64 output tiles x batch slices into slabs,
//                        + tower_wgrad_finish_kernel: the slices of every tile summed in slice order (deterministic)
// Why: at Criteo-Kaggle shapes (BASELINE.json configs[1]: batch 2048, widths 13-512-256-64-16 / 367-512-256-1) a training step is
// ~45 dependent kernels of 4-25 us, most of it dependency latency: the per-layer GEMMs (64 x 64 tiles, split-k slabs + a reduction
// kernel per weight gradient) occupy a fraction of the chip for a few microseconds each (profiles/round5/step_trace_kaggle_graph.txt).
// Large batches keep the per-layer LDS-DMA GEMMs of gemm.hip: here every workgroup re-reads all weights, which only pays while
// (M / 16) x sum(N_l x K_l) x 4 bytes of L2 traffic is small.
//
// MFMA use (v_mfma_f32_16x16x4_f32; lane l: li = l & 15, g = l >> 4; A[i = li][k = g], B[k = g][j = li], D[i = 4g + r][j = li]):
// a lane loads FOUR consecutive floats of an operand row with one 16-byte load and feeds them to four MFMAs — the k index of a
// product only has to agree between A and B, so MFMA c of a 16-wide k-step multiplies the k values {4g + c}.  Where the four floats
// run along an OUTPUT dimension instead (backward: W rows, weight gradient: both operands) they select four different output
// columns / rows: output j of MFMA c stands for column 4j + c, which makes a lane's four results of consecutive c a float4 again.
#include "common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int TW_MAXL = DLRM_TOWER_MAX_LAYERS;
constexpr int TW_ROWS = 16;            // batch rows per workgroup (one MFMA row block)
constexpr int TW_NW = 16;              // forward / backward: waves per workgroup (1024 threads: four waves per SIMD hide the L2 latency of the weight loads)
constexpr int TW_THREADS = 64 * TW_NW;
constexpr int TW_TPW = 2;              // forward / backward: output column tiles (16 wide) a wave advances together
constexpr int TW_KU = 2;               // forward / backward: 16-wide reduction steps per chunk (the next chunk's operands are in flight meanwhile)
constexpr int TW_KUB = 1;              // ... of the backward loop (four 4-byte loads per step and tile: one step per chunk keeps it inside 128 registers)
constexpr int TW_WU = 4;               // weight gradient: 4-row groups per chunk
constexpr int TW_UNR = 8;              // pipeline stages per loop body
constexpr int TW_UNRB = 4;             // ... of the backward loop (register budget of 1024 threads)

// one pipeline stage of the hand-pipelined loops: the loads of the NEXT chunk first, back to back, then the products of the current one.
// Left alone the scheduler sinks every load to just before its first use in the next stage, i.e. one exposed L2 round trip per load.
#define TW_STAGE_ORDER(n_vmem, n_ds, n_mfma)                      \
    do {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x020, (n_vmem), 0); \
        __builtin_amdgcn_sched_group_barrier(0x100, (n_ds), 0);   \
        __builtin_amdgcn_sched_group_barrier(0x008, (n_mfma), 0); \
    } while (0)

struct TowerArgs {
    int L, M;
    int width[TW_MAXL + 1];            // width[0] = input width of the tower, width[l + 1] = outputs of layer l
    int act[TW_MAXL];
    const float* W[TW_MAXL]; long long ldw[TW_MAXL];
    const float* bias[TW_MAXL];
    const float* X; long long ldx;     // tower input (forward) / incoming gradient dY (backward)
    float* Y[TW_MAXL]; long long ldy[TW_MAXL];      // layer outputs: written by forward, read by backward
    float* dZ[TW_MAXL]; long long lddz[TW_MAXL];    // backward: dL/dz of every layer, written
    float* dX; long long lddx;         // backward: gradient of the tower input (nullable)
    int ld_lds;                        // row pitch (floats) of the LDS activation buffers
    int last_act_applied;              // backward: dY already is dL/dz of the last layer
};

__device__ __forceinline__ float tw_act(float v, int act) {
    if (act == DLRM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DLRM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
__device__ __forceinline__ float tw_act_grad(float g, float y, int act) {       // through the OUTPUT y, as dlrm_act_bwd
    if (act == DLRM_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DLRM_ACT_SIGMOID) return g * ((1.f - y) * y);
    return g;
}
__device__ __forceinline__ bool tw_vec_ok(const void* p, long long ld, int cols) {
    return (((uintptr_t)p) & 15u) == 0 && ld % 4 == 0 && cols % 4 == 0;
}
// Operand loads of the product loops: four consecutive floats row[c .. c+3] of a [rows, cols] matrix (VEC: 16-byte rows, cols % 4 == 0, so a
// quad is in or out as a whole).  An out-of-range request is CLAMPED to a valid element (row 0 / column 0) and comes back as whatever is
// there — no branch and no select: a predicated load costs the loop a branch per load, and a select on the loaded value makes the loop wait
// for the load where it is issued instead of where it is used.  Every caller pairs a clamped operand with a ZERO in the other operand (the
// LDS activation buffers are zero beyond a layer's width and beyond the batch) or discards the affected outputs; weights are finite.
template <bool VEC>
__device__ __forceinline__ float4 tw_load4(const float* __restrict__ p, long long ld, int r, int c, int rows, int cols) {
    const float* q = p + (r < rows ? (long long)r * ld : 0);
    if (VEC) return *(const float4*)(q + (c + 3 < cols ? c : 0));
    return make_float4(q[c < cols ? c : 0], q[c + 1 < cols ? c + 1 : 0], q[c + 2 < cols ? c + 2 : 0], q[c + 3 < cols ? c + 3 : 0]);
}
template <bool VEC>
__device__ __forceinline__ void tw_store4(float* __restrict__ p, long long ld, int r, int c, int rows, int cols, float4 v) {
    if (r >= rows) return;
    float* q = p + (long long)r * ld + c;
    if (VEC) { if (c + 3 < cols) *(float4*)q = v; return; }
    if (c < cols) q[0] = v.x;
    if (c + 1 < cols) q[1] = v.y;
    if (c + 2 < cols) q[2] = v.z;
    if (c + 3 < cols) q[3] = v.w;
}

// ------------------------------------------------------------------------------------------------ forward
// one layer: nxt[16][N] = act(cur[16][K] . W[N, K]^T + b), also to Y.  Waves take column tiles t = wave*TPW + 4*TPW*pass + j.
template <bool VEC>
__device__ __forceinline__ void tw_fwd_layer(const TowerArgs& a, int l, const float* cur, float* nxt, int m0, int lane, int wave) {
    const int li = lane & 15, g = lane >> 4;
    const int K = a.width[l], N = a.width[l + 1];
    const int Kp = (K + 15) & ~15, ntiles = (N + 15) >> 4;
    const int LD = a.ld_lds;
    const float* __restrict__ W = a.W[l];
    const long long ldw = a.ldw[l];
    // a pass covers NW * TPW tiles: wave w takes tiles tb + w, tb + w + NW, ... (narrow layers still spread over the waves)
    for (int tb = 0; tb < ntiles; tb += TW_NW * TW_TPW) {
        const int t0 = tb + wave;                        // tile j of this wave: t0 + NW * j
        if (t0 >= ntiles) break;
        floatx4 acc[TW_TPW];
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        // one wave per SIMD and operands that come from L2: the k loop runs in chunks of TW_KU 16-wide steps and the NEXT chunk's
        // loads are issued before the current chunk's products (explicit double buffering; the compiler does not pipeline this loop)
        float4 a4[2][TW_KU], b4[2][TW_KU][TW_TPW];
        bool kv[2][TW_KU];                                   // 16-wide step inside the reduction (a chunk may end past Kp: zeroed AT USE)
        auto load = [&](int k0, float4 (&av)[TW_KU], float4 (&bv)[TW_KU][TW_TPW], bool (&ok)[TW_KU]) {
#pragma unroll
            for (int u = 0; u < TW_KU; ++u) {
                const int k = k0 + 16 * u;
                // (k in [K, Kp): the activation columns read from LDS are zeros — every producer of a buffer zero-fills up to the next
                // multiple of 16 — so whatever the clamped weight load returns adds nothing; tiles past the layer's last one compute
                // columns that the epilogue drops)
                ok[u] = k < Kp;
                av[u] = *(const float4*)(cur + li * LD + (ok[u] ? k : 0) + 4 * g);
#pragma unroll
                for (int j = 0; j < TW_TPW; ++j)
                    bv[u][j] = tw_load4<VEC>(W, ldw, (t0 + TW_NW * j) * 16 + li, k + 4 * g, N, K);
            }
        };
        auto mma = [&](const float4 (&av)[TW_KU], const float4 (&bv)[TW_KU][TW_TPW], const bool (&ok)[TW_KU]) {
#pragma unroll
            for (int u = 0; u < TW_KU; ++u) {
                const float ax = ok[u] ? av[u].x : 0.f, ay = ok[u] ? av[u].y : 0.f, az = ok[u] ? av[u].z : 0.f, aw = ok[u] ? av[u].w : 0.f;
#pragma unroll
                for (int j = 0; j < TW_TPW; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bv[u][j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, bv[u][j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bv[u][j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bv[u][j].w, acc[j], 0, 0, 0);
                }
            }
        };
        // TW_UNR stages per loop body, the pipeline drained at the body's end: the compiler's wait-count insertion is exact inside a body
        // but conservative across a loop back-edge (it would make every stage wait for the loads it has just issued)
        constexpr int CH = 16 * TW_KU;
        for (int kb = 0; kb < Kp; kb += TW_UNR * CH) {
            load(kb, a4[0], b4[0], kv[0]);
#pragma unroll
            for (int u = 0; u < TW_UNR; ++u) {
                // the next chunk is requested UNCONDITIONALLY (past the end: clamped addresses, dropped): a load under a condition makes
                // the wait-count insertion merge "issued" and "not issued" at the join and wait for the newest loads as well
                if (u + 1 < TW_UNR) load(kb + (u + 1) * CH, a4[(u + 1) & 1], b4[(u + 1) & 1], kv[(u + 1) & 1]);
                mma(a4[u & 1], b4[u & 1], kv[u & 1]);
                TW_STAGE_ORDER((VEC ? 1 : 4) * TW_KU * TW_TPW, TW_KU, 4 * TW_KU * TW_TPW);
                if (kb + (u + 1) * CH >= Kp) break;
            }
        }
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) {
            if (t0 + TW_NW * j >= ntiles) continue;
            const int n = (t0 + TW_NW * j) * 16 + li;
            const float b = (n < N && a.bias[l]) ? a.bias[l][n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const float v = n < N ? tw_act(acc[j][r] + b, a.act[l]) : 0.f;     // columns N .. 16*ntiles-1: the next layer's k padding
                nxt[row * LD + n] = v;
                if (n < N && m0 + row < a.M) a.Y[l][(long long)(m0 + row) * a.ldy[l] + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(TW_THREADS) void tower_fwd_kernel(TowerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * TW_ROWS;
    const int LD = a.ld_lds;
    float* cur = lds;
    float* nxt = lds + TW_ROWS * LD;
    {   // the tower's input rows, zero beyond the batch and up to the next multiple of 16 columns
        const int K0 = a.width[0], K0p = (K0 + 15) & ~15;
        for (int e = threadIdx.x; e < TW_ROWS * K0p; e += TW_THREADS) {
            const int r = e / K0p, c = e - r * K0p;
            cur[r * LD + c] = (m0 + r < a.M && c < K0) ? a.X[(long long)(m0 + r) * a.ldx + c] : 0.f;
        }
    }
    __syncthreads();
    for (int l = 0; l < a.L; ++l) {
        if (tw_vec_ok(a.W[l], a.ldw[l], a.width[l])) tw_fwd_layer<true>(a, l, cur, nxt, m0, lane, wave);
        else tw_fwd_layer<false>(a, l, cur, nxt, m0, lane, wave);
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
}

// ------------------------------------------------------------------------------------------------ backward (data gradients)
// one layer: dA[16][K] = cur[16][N] (= dZ_l) . W[N, K]; then dZ_{l-1} = dA * act'_{l-1}(Y_{l-1}) -> nxt and dZ[l-1]   (l > 0)
//                                                       or dX = dA                                              (l == 0)
// Waves take 16-column tiles of dA.  A = dZ rows (one 16-byte LDS read feeds the four MFMAs of a 16-wide step, MFMA s multiplying
// the reduction indices n0 + 4g + s), B[k = g][j = li] = W[n0 + 4g + s][c + li]: the reduction runs over ROWS of W, so a lane's
// four values are four 4-byte loads (16 lanes cover 64 contiguous bytes of a row; the neighbouring tile — the next wave's — takes the
// other half of the cache line).
__device__ __forceinline__ float tw_load1(const float* __restrict__ p, long long ld, int r, int c, int rows, int cols) {       // (clamped like tw_load4)
    return p[(r < rows ? (long long)r * ld : 0) + (c < cols ? c : 0)];
}
__device__ __forceinline__ void tw_bwd_layer(const TowerArgs& a, int l, const float* cur, float* nxt, int m0, int lane, int wave) {
    const int li = lane & 15, g = lane >> 4;
    const int K = a.width[l], N = a.width[l + 1];
    const int Np = (N + 15) & ~15, ntiles = (K + 15) >> 4;
    const int LD = a.ld_lds;
    const float* __restrict__ W = a.W[l];
    const long long ldw = a.ldw[l];
    for (int tb = 0; tb < ntiles; tb += TW_NW * TW_TPW) {
        const int t0 = tb + wave;
        if (t0 >= ntiles) break;
        floatx4 acc[TW_TPW];
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        float4 a4[2][TW_KUB];
        float b1[2][TW_KUB][TW_TPW][4];                   // (double buffered like the forward loop)
        auto load = [&](int n0, float4 (&av)[TW_KUB], float (&bv)[TW_KUB][TW_TPW][4]) {
#pragma unroll
            for (int u = 0; u < TW_KUB; ++u) {
                const int n = n0 + 16 * u;
                av[u] = *(const float4*)(cur + li * LD + (n < Np ? n : 0) + 4 * g);          // (zeros for n in [N, Np): see the forward loop; past Np: a chunk that is dropped)
#pragma unroll
                for (int j = 0; j < TW_TPW; ++j)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
                        bv[u][j][s_] = tw_load1(W, ldw, n + 4 * g + s_, (t0 + TW_NW * j) * 16 + li, N, K);
            }
        };
        auto mma = [&](const float4 (&av)[TW_KUB], const float (&bv)[TW_KUB][TW_TPW][4]) {
#pragma unroll
            for (int u = 0; u < TW_KUB; ++u) {
                const float as[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
                for (int j = 0; j < TW_TPW; ++j) {
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s_], bv[u][j][s_], acc[j], 0, 0, 0);
                }
            }
        };
        constexpr int CH = 16 * TW_KUB;                      // (TW_UNRB stages per body, drained at its end: see the forward loop)
        for (int nb = 0; nb < Np; nb += TW_UNRB * CH) {
            load(nb, a4[0], b1[0]);
#pragma unroll
            for (int u = 0; u < TW_UNRB; ++u) {
                if (u + 1 < TW_UNRB) load(nb + (u + 1) * CH, a4[(u + 1) & 1], b1[(u + 1) & 1]);      // (unconditional: see the forward loop)
                mma(a4[u & 1], b1[u & 1]);
                TW_STAGE_ORDER(4 * TW_KUB * TW_TPW, TW_KUB, 4 * TW_KUB * TW_TPW);
                if (nb + (u + 1) * CH >= Np) break;
            }
        }
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) {
            if (t0 + TW_NW * j >= ntiles) continue;
            const int c = (t0 + TW_NW * j) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                float v = acc[j][r];
                if (l > 0) {
                    const bool in = c < K && m0 + row < a.M;
                    v = in ? tw_act_grad(v, a.Y[l - 1][(long long)(m0 + row) * a.ldy[l - 1] + c], a.act[l - 1]) : 0.f;
                    nxt[row * LD + c] = v;                        // (zeros beyond K: the k padding of the next product)
                    if (in) a.dZ[l - 1][(long long)(m0 + row) * a.lddz[l - 1] + c] = v;
                } else if (c < K && m0 + row < a.M) {
                    a.dX[(long long)(m0 + row) * a.lddx + c] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(TW_THREADS) void tower_bwd_kernel(TowerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * TW_ROWS;
    const int LD = a.ld_lds;
    float* cur = lds;
    float* nxt = lds + TW_ROWS * LD;
    {   // dZ of the last layer from the incoming gradient (a.X) and the layer's output
        const int l = a.L - 1, N = a.width[a.L], Np = (N + 15) & ~15;
        for (int e = threadIdx.x; e < TW_ROWS * Np; e += TW_THREADS) {
            const int r = e / Np, c = e - r * Np;
            float v = 0.f;
            if (m0 + r < a.M && c < N) {
                v = a.X[(long long)(m0 + r) * a.ldx + c];
                if (!a.last_act_applied) v = tw_act_grad(v, a.Y[l][(long long)(m0 + r) * a.ldy[l] + c], a.act[l]);
                a.dZ[l][(long long)(m0 + r) * a.lddz[l] + c] = v;
            }
            cur[r * LD + c] = v;
        }
    }
    __syncthreads();
    const int stop = a.dX ? 0 : 1;
    for (int l = a.L - 1; l >= stop; --l) {
        tw_bwd_layer(a, l, cur, nxt, m0, lane, wave);
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
struct TowerWgradArgs {
    int L, M, S;                       // layers, batch rows, batch slices per tile
    int rows_per_slice;                // multiple of 16
    int N[TW_MAXL], K[TW_MAXL], kstore[TW_MAXL];     // dW_l is [N, kstore] (kstore <= K: the input's alignment padding is dropped)
    int tile0[TW_MAXL + 1];            // prefix of the layers' 64 x 64 tile counts
    const float* dZ[TW_MAXL]; long long lddz[TW_MAXL];
    const float* In[TW_MAXL]; long long ldin[TW_MAXL];
    float* dW[TW_MAXL]; long long lddw[TW_MAXL];
    float* db[TW_MAXL];                // nullable
    float* slabs;                      // [tiles][S][64*64 + 64]
};
constexpr int TW_SLAB = 64 * 64 + 64;

template <bool VZ, bool VI>
__device__ __forceinline__ void tw_wgrad_accumulate(const TowerWgradArgs& a, int l, int n0, int k0, int m_begin, int m_end, int lane, int wave,
                                                    floatx4 (&acc)[4][4], float (&bs)[4]) {
    const int li = lane & 15, g = lane >> 4;
    const float* __restrict__ dZ = a.dZ[l];
    const float* __restrict__ In = a.In[l];
    const int N = a.N[l], K = a.K[l];
    // wave w takes the 4-row groups w, w + 4, ... of the slice; lane group g the row g of a group.  Chunks of TW_WU groups, the next
    // chunk's rows requested before the current chunk's products.
    float4 z4[2][TW_WU], x4[2][TW_WU];
    bool rv[2][TW_WU];                                          // row inside the slice (else its — clamped — dZ values are zeroed AT USE)
    auto load = [&](int m, float4 (&zv)[TW_WU], float4 (&xv)[TW_WU], bool (&ok)[TW_WU]) {
#pragma unroll
        for (int u = 0; u < TW_WU; ++u) {
            const int row = m + 16 * u + g;
            ok[u] = row < m_end;
            zv[u] = tw_load4<VZ>(dZ, a.lddz[l], row, n0 + 4 * li, m_end, N);
            xv[u] = tw_load4<VI>(In, a.ldin[l], row, k0 + 4 * li, m_end, K);
        }
    };
    auto mma = [&](const float4 (&zv)[TW_WU], const float4 (&xv)[TW_WU], const bool (&ok)[TW_WU]) {
#pragma unroll
        for (int u = 0; u < TW_WU; ++u) {
            const float zs[4] = {ok[u] ? zv[u].x : 0.f, ok[u] ? zv[u].y : 0.f, ok[u] ? zv[u].z : 0.f, ok[u] ? zv[u].w : 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[c], xv[u].x, acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[c], xv[u].y, acc[c][1], 0, 0, 0);
                acc[c][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[c], xv[u].z, acc[c][2], 0, 0, 0);
                acc[c][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[c], xv[u].w, acc[c][3], 0, 0, 0);
                bs[c] += zs[c];
            }
        }
    };
    const int mb = m_begin + 4 * wave;
    constexpr int CH = 16 * TW_WU;                           // (TW_UNR stages per body, drained at its end: see the forward loop)
    for (int m = mb; m < m_end; m += TW_UNR * CH) {
        load(m, z4[0], x4[0], rv[0]);
#pragma unroll
        for (int u = 0; u < TW_UNR; ++u) {
            if (u + 1 < TW_UNR) load(m + (u + 1) * CH, z4[(u + 1) & 1], x4[(u + 1) & 1], rv[(u + 1) & 1]);      // (unconditional: see the forward loop)
            mma(z4[u & 1], x4[u & 1], rv[u & 1]);
            TW_STAGE_ORDER(((VZ ? 1 : 4) + (VI ? 1 : 4)) * TW_WU, 0, 16 * TW_WU);
            if (m + (u + 1) * CH >= m_end) break;
        }
    }
}

// grid = tiles * S.  Workgroup (tile, s): partial of the 64 x 64 tile over batch slice s — the four waves' partials summed through LDS in
// wave order — into its slab (S == 1: straight into dW / db).  tower_wgrad_finish_kernel then adds a tile's slabs in slice order.
// (One launch with a "last workgroup of the tile reduces" ticket was measured first: the device-scope fences it needs write back and
// invalidate the L2 of every XCD — 134 us for a launch whose products take 5, profiles/round5/step_trace_kaggle_graph_towers_v1.txt.)
__device__ __forceinline__ void tw_store_tile_row(const TowerWgradArgs& a, int l, int n, int k, float4 v) {
    if (n >= a.N[l]) return;
    const int KS = a.kstore[l];
    float* q = a.dW[l] + (long long)n * a.lddw[l] + k;
    if (k + 3 < KS && (((uintptr_t)q) & 15u) == 0) { *(float4*)q = v; return; }
    if (k < KS) q[0] = v.x;
    if (k + 1 < KS) q[1] = v.y;
    if (k + 2 < KS) q[2] = v.z;
    if (k + 3 < KS) q[3] = v.w;
}

__global__ __launch_bounds__(256) void tower_wgrad_kernel(TowerWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float red[3][TW_SLAB];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x / a.S, s = blockIdx.x - tile * a.S;
    int l = 0;
    while (l + 1 < a.L && tile >= a.tile0[l + 1]) ++l;
    const int tk_n = (a.K[l] + 63) >> 6;
    const int tl = tile - a.tile0[l];
    const int n0 = (tl / tk_n) * 64, k0 = (tl - (tl / tk_n) * tk_n) * 64;
    const int m_begin = s * a.rows_per_slice;
    const int m_end = (m_begin + a.rows_per_slice < a.M) ? m_begin + a.rows_per_slice : a.M;

    floatx4 acc[4][4];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = (floatx4){0.f, 0.f, 0.f, 0.f};
    if (m_begin < m_end) {
        const bool vz = tw_vec_ok(a.dZ[l], a.lddz[l], a.N[l]), vi = tw_vec_ok(a.In[l], a.ldin[l], a.K[l]);
        if (vz && vi) tw_wgrad_accumulate<true, true>(a, l, n0, k0, m_begin, m_end, lane, wave, acc, bs);
        else if (vz) tw_wgrad_accumulate<true, false>(a, l, n0, k0, m_begin, m_end, lane, wave, acc, bs);
        else if (vi) tw_wgrad_accumulate<false, true>(a, l, n0, k0, m_begin, m_end, lane, wave, acc, bs);
        else tw_wgrad_accumulate<false, false>(a, l, n0, k0, m_begin, m_end, lane, wave, acc, bs);
    }
    // lane (li, g) holds acc[c][d][r] = dW[n0 + 4*(4g + r) + c][k0 + 4*li + d]; slab layout [64 n][64 k] then [64] bias sums.
    // The bias sums of a lane cover ITS row (g) of every group: the four lane groups are folded first (fixed order).
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = bs[c];
        const float v0 = __shfl(v, li, 64), v1 = __shfl(v, li + 16, 64), v2 = __shfl(v, li + 32, 64), v3 = __shfl(v, li + 48, 64);
        bs[c] = ((v0 + v1) + v2) + v3;
    }
    auto put = [&](float* dst) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *(float4*)(dst + (4 * (4 * g + r) + c) * 64 + 4 * li) = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[64 * 64 + 4 * li + c] = bs[c];
        }
    };
    // waves 1..3 park their partials, wave 0 adds them in wave order
    if (wave > 0) put(red[wave - 1]);
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 o = *(const float4*)(red[w] + (4 * (4 * g + r) + c) * 64 + 4 * li);
                acc[c][0][r] += o.x; acc[c][1][r] += o.y; acc[c][2][r] += o.z; acc[c][3][r] += o.w;
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) bs[c] += red[w][64 * 64 + 4 * li + c];
    }
    if (a.S > 1) { put(a.slabs + ((long long)tile * a.S + s) * TW_SLAB); return; }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            tw_store_tile_row(a, l, n0 + 4 * (4 * g + r) + c, k0 + 4 * li, make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]));
    if (g == 0 && k0 == 0 && a.db[l]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (n0 + 4 * li + c < a.N[l]) a.db[l][n0 + 4 * li + c] = bs[c];
    }
}

// grid = tiles * 4: workgroup (tile, q) adds the S slabs of rows 16q .. 16q + 15 of the tile in slice order and writes dW (q == 0 also db)
__global__ __launch_bounds__(256) void tower_wgrad_finish_kernel(TowerWgradArgs a) {
    const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;
    int l = 0;
    while (l + 1 < a.L && tile >= a.tile0[l + 1]) ++l;
    const int tk_n = (a.K[l] + 63) >> 6;
    const int tl = tile - a.tile0[l];
    const int n0 = (tl / tk_n) * 64, k0 = (tl - (tl / tk_n) * tk_n) * 64;
    const float* base = a.slabs + (long long)tile * a.S * TW_SLAB;
    const int nr = 16 * q + (threadIdx.x >> 4), kq = (threadIdx.x & 15) * 4;
    float4 v = *(const float4*)(base + nr * 64 + kq);
    for (int t = 1; t < a.S; ++t) {
        const float4 o = *(const float4*)(base + (long long)t * TW_SLAB + nr * 64 + kq);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    tw_store_tile_row(a, l, n0 + nr, k0 + kq, v);
    if (q == 0 && k0 == 0 && a.db[l] && threadIdx.x < 64) {
        float b = base[64 * 64 + threadIdx.x];
        for (int t = 1; t < a.S; ++t) b += base[(long long)t * TW_SLAB + 64 * 64 + threadIdx.x];
        if (n0 + (int)threadIdx.x < a.N[l]) a.db[l][n0 + threadIdx.x] = b;
    }
}

static int tw_check(int64_t M, int L, const int* widths) {
    if (M <= 0 || M > 0x7fffffff - 64 || L <= 0 || !widths) return DLRM_E_ARG;
    if (L > TW_MAXL) return DLRM_E_RANGE;
    for (int l = 0; l <= L; ++l) {
        if (widths[l] <= 0) return DLRM_E_ARG;
        if (widths[l] > DLRM_TOWER_MAX_WIDTH) return DLRM_E_RANGE;
    }
    return 0;
}
static int tw_pitch(int L, const int* widths) {
    int w = 0;
    for (int l = 0; l <= L; ++l) if (widths[l] > w) w = widths[l];
    return ((w + 63) & ~63) + 4;          // whole 64-column blocks (backward writes them) + 4: rows start 4 banks apart
}

}  // namespace

extern "C" int dlrm_tower_fwd(int64_t M, int L, const int* widths, const int* acts, const float* X, int64_t ldx, const void* const* W_host,
                              const int64_t* ldw_host, const void* const* bias_host, void* const* Y_host, const int64_t* ldy_host, void* stream) {
    int rc = tw_check(M, L, widths);
    if (rc) return rc;
    if (!acts || !X || !W_host || !ldw_host || !bias_host || !Y_host || !ldy_host || ldx < widths[0]) return DLRM_E_ARG;
    TowerArgs a = {};
    a.L = L; a.M = (int)M; a.X = X; a.ldx = ldx; a.ld_lds = tw_pitch(L, widths);
    a.width[0] = widths[0];
    for (int l = 0; l < L; ++l) {
        if (!W_host[l] || !Y_host[l] || ldw_host[l] < widths[l] || ldy_host[l] < widths[l + 1] || acts[l] < 0 || acts[l] > 2) return DLRM_E_ARG;
        a.width[l + 1] = widths[l + 1]; a.act[l] = acts[l];
        a.W[l] = (const float*)W_host[l]; a.ldw[l] = ldw_host[l]; a.bias[l] = (const float*)bias_host[l];
        a.Y[l] = (float*)Y_host[l]; a.ldy[l] = ldy_host[l];
    }
    const size_t lds = (size_t)2 * TW_ROWS * a.ld_lds * sizeof(float);
    (void)hipFuncSetAttribute((const void*)tower_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tower_fwd_kernel, dim3((unsigned)((M + TW_ROWS - 1) / TW_ROWS)), dim3(TW_THREADS), lds, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_tower_bwd(int64_t M, int L, const int* widths, const int* acts, const float* dY, int64_t lddy, int last_act_applied,
                              const void* const* W_host, const int64_t* ldw_host, const void* const* Y_host, const int64_t* ldy_host,
                              void* const* dZ_host, const int64_t* lddz_host, float* dX, int64_t lddx, void* stream) {
    int rc = tw_check(M, L, widths);
    if (rc) return rc;
    if (!acts || !dY || !W_host || !ldw_host || !Y_host || !ldy_host || !dZ_host || !lddz_host || lddy < widths[L] || (dX && lddx < widths[0]))
        return DLRM_E_ARG;
    TowerArgs a = {};
    a.L = L; a.M = (int)M; a.X = dY; a.ldx = lddy; a.ld_lds = tw_pitch(L, widths); a.dX = dX; a.lddx = lddx;
    a.last_act_applied = last_act_applied ? 1 : 0;
    a.width[0] = widths[0];
    for (int l = 0; l < L; ++l) {
        if (!W_host[l] || !Y_host[l] || !dZ_host[l] || ldw_host[l] < widths[l] || ldy_host[l] < widths[l + 1] || lddz_host[l] < widths[l + 1] ||
            acts[l] < 0 || acts[l] > 2) return DLRM_E_ARG;
        a.width[l + 1] = widths[l + 1]; a.act[l] = acts[l];
        a.W[l] = (const float*)W_host[l]; a.ldw[l] = ldw_host[l];
        a.Y[l] = (float*)Y_host[l]; a.ldy[l] = ldy_host[l];
        a.dZ[l] = (float*)dZ_host[l]; a.lddz[l] = lddz_host[l];
    }
    const size_t lds = (size_t)2 * TW_ROWS * a.ld_lds * sizeof(float);
    (void)hipFuncSetAttribute((const void*)tower_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tower_bwd_kernel, dim3((unsigned)((M + TW_ROWS - 1) / TW_ROWS)), dim3(TW_THREADS), lds, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// batch slices per tile: enough workgroups for four per CU, slices of at least 64 rows
static void tw_wgrad_plan(int64_t M, int L, const int* N, const int* K, int* tiles_out, int* S_out, int* rows_out) {
    int tiles = 0;
    for (int l = 0; l < L; ++l) tiles += ((N[l] + 63) / 64) * ((K[l] + 63) / 64);
    int S = (1024 + tiles - 1) / tiles;
    const int64_t maxS = (M + 63) / 64;
    if (S > maxS) S = (int)maxS;
    if (S < 1) S = 1;
    int64_t rows = (M + S - 1) / S;
    rows = (rows + 15) & ~(int64_t)15;
    *tiles_out = tiles; *S_out = (int)((M + rows - 1) / rows); *rows_out = (int)rows;
}

extern "C" int64_t dlrm_tower_wgrad_workspace_bytes(int64_t M, int L, const int* widths) {
    if (tw_check(M, L, widths)) return 0;
    int tiles, S, rows;
    tw_wgrad_plan(M, L, widths + 1, widths, &tiles, &S, &rows);
    return (int64_t)tiles * S * TW_SLAB * (int64_t)sizeof(float) + 256;
}

extern "C" int dlrm_tower_wgrad(int64_t M, int L, const int* widths, const int* kstore, const void* const* dZ_host, const int64_t* lddz_host,
                                const void* const* In_host, const int64_t* ldin_host, void* const* dW_host, const int64_t* lddw_host,
                                void* const* db_host, void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = tw_check(M, L, widths);
    if (rc) return rc;
    if (!kstore || !dZ_host || !lddz_host || !In_host || !ldin_host || !dW_host || !lddw_host || !db_host || !workspace) return DLRM_E_ARG;
    if (workspace_bytes < dlrm_tower_wgrad_workspace_bytes(M, L, widths) || !dlrm_aligned16(workspace)) return DLRM_E_ARG;
    TowerWgradArgs a = {};
    int tiles;
    a.L = L; a.M = (int)M;
    for (int l = 0; l < L; ++l) {
        a.N[l] = widths[l + 1]; a.K[l] = widths[l]; a.kstore[l] = kstore[l];
        if (kstore[l] <= 0 || kstore[l] > widths[l] || !dZ_host[l] || !In_host[l] || !dW_host[l] || lddz_host[l] < widths[l + 1] ||
            ldin_host[l] < widths[l] || lddw_host[l] < kstore[l]) return DLRM_E_ARG;
        a.dZ[l] = (const float*)dZ_host[l]; a.lddz[l] = lddz_host[l];
        a.In[l] = (const float*)In_host[l]; a.ldin[l] = ldin_host[l];
        a.dW[l] = (float*)dW_host[l]; a.lddw[l] = lddw_host[l]; a.db[l] = (float*)db_host[l];
    }
    tw_wgrad_plan(M, L, a.N, a.K, &tiles, &a.S, &a.rows_per_slice);
    a.tile0[0] = 0;
    for (int l = 0; l < L; ++l) a.tile0[l + 1] = a.tile0[l] + ((a.N[l] + 63) / 64) * ((a.K[l] + 63) / 64);
    for (int l = L; l < TW_MAXL; ++l) a.tile0[l + 1] = a.tile0[L];
    a.slabs = (float*)workspace;
    hipLaunchKernelGGL(tower_wgrad_kernel, dim3((unsigned)(tiles * a.S)), dim3(256), 0, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    if (a.S > 1) {
        hipLaunchKernelGGL(tower_wgrad_finish_kernel, dim3((unsigned)(tiles * 4)), dim3(256), 0, (hipStream_t)stream, a);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}
