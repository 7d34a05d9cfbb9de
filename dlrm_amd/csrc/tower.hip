// tower.hip — an MLP tower (nn.Sequential(Linear, act, Linear, act, ...), dlrm_s_pytorch.py:208-246, 399-405) for SMALL batches as
// four launches instead of one GEMM launch per layer and direction (plus a reduction launch per weight gradient):
//   tower_fwd_kernel           all layers forward: a workgroup (1024 threads) owns 16 batch rows whose activations stay in LDS from layer
//                              to layer (each is also written out: the backward pass reads them); the weights stream through LDS panels
//   tower_bwd_kernel           the data-gradient chain dZ_L -> dZ_{L-1} -> ... (-> dX), same ownership; every layer's dZ is written out
//   tower_wgrad_kernel         the weight / bias gradients of ALL layers in one launch: 64 x 64 output tiles x batch slices into slabs
//   tower_wgrad_finish_kernel  the slices of every tile summed in slice order (deterministic)
// Why: at Criteo-Kaggle shapes (BASELINE.json configs[1]: batch 2048, widths 13-512-256-64-16 / 367-512-256-1) a training step is
// ~45 dependent kernels of 4-25 us: the per-layer GEMMs (64 x 64 tiles, split-k slabs + a reduction kernel per weight gradient) occupy
// a fraction of the chip for a few microseconds each, and a replayed HIP graph pays per node (profiles/round5/step_trace_kaggle_graph.txt).
// With all of these kernels the step is 19 kernels.  What they are NOT is fast per kernel: a batch of 2048 rows is 128 workgroups — half
// the chip — and every workgroup streams ALL weights for its 16 rows (no reuse beyond one MFMA row block), so the forward / backward
// launches run at about a third of the MFMA rate.  In the BACKWARD pass they replace three launches per layer and come out even in kernel
// time, in the forward pass one launch per layer and lose to the per-layer GEMMs (which spread a layer over all 256 CUs): the default is
// per-layer forward + tower backward, 0.388 -> 0.320 ms per graphed step (towers everywhere: 0.358; profiles/round5/kaggle_towers.md has
// the five versions that were measured).  Large batches keep the per-layer LDS-DMA GEMMs of gemm.hip altogether: (M / 16) x
// sum(N_l x K_l) x 4 bytes of weight traffic only pays while it is small (functional._tower_applies).
//
// MFMA use (v_mfma_f32_16x16x4_f32; lane l: li = l & 15, g = l >> 4; A[i = li][k = g], B[k = g][j = li], D[i = 4g + r][j = li]):
// a lane loads FOUR consecutive floats of an operand row with one 16-byte load and feeds them to four MFMAs — the k index of a
// product only has to agree between A and B, so MFMA c of a 16-wide k-step multiplies the k values {4g + c}.  Where the four floats
// run along an OUTPUT dimension instead (weight gradient: both operands) they select four different output rows / columns: output j
// of MFMA c stands for column 4j + c, which makes a lane's four results of consecutive c a float4 again.
#include "common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int TW_MAXL = DLRM_TOWER_MAX_LAYERS;
constexpr int TW_ROWS = 16;            // batch rows per workgroup (one MFMA row block)
constexpr int TW_NW = 16;              // forward / backward: waves per workgroup (1024 threads)
constexpr int TW_THREADS = 64 * TW_NW;
constexpr int TW_TPW = 2;              // forward / backward: output column tiles (16 wide) per wave and pass
constexpr int TW_WU = 4;               // weight gradient: 4-row groups per chunk
constexpr int TW_UNR = 8;              // weight gradient: pipeline stages per loop body

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
// ------------------------------------------------------------------------------------------------ forward / backward products
// Weights reach the MFMAs through LDS: per chunk of the reduction all 1024 threads fetch a [rows x 32 k] (forward) / [32 n x K] (backward)
// panel of W with full-line, row-contiguous 16-byte loads into registers WHILE earlier panels are multiplied, then park it in LDS
// between two barriers.  (Fragments loaded straight from global memory were measured first — a tile's B operand is 16 rows x 16 bytes
// per instruction, i.e. 16 half-used cache lines — with the same result within 20 % either way: profiles/round5/kaggle_towers.md.)
// Panel elements outside W are stored as ZEROS, so the activation buffers only have to hold finite values wherever they are read
// (they are cleared when the kernel starts).
constexpr int TW_CK = 32;                          // reduction elements per chunk
constexpr int TW_PASS = 512;                       // forward: output columns per pass (32 tiles: two per wave)
constexpr int TW_WLD_F = TW_CK + 4;                // forward panel [512][36]: rows 144 bytes apart (16-byte fragments, conflict free)
constexpr int TW_WLD_B = DLRM_TOWER_MAX_WIDTH + 4; // backward panel [32][516]: four k-groups two banks apart
constexpr int TW_WST_FLOATS = (TW_PASS * TW_WLD_F > TW_CK * TW_WLD_B) ? TW_PASS * TW_WLD_F : TW_CK * TW_WLD_B;

// four floats W[r][c .. c+3] (zeros outside [rows, cols]); clamped address + select: no branch (see tw_load4)
template <bool VEC>
__device__ __forceinline__ float4 tw_panel_quad(const float* __restrict__ W, long long ldw, int r, int c, int rows, int cols) {
    const bool rok = r < rows;
    const float4 v = tw_load4<VEC>(W, ldw, r, c, rows, cols);
    return make_float4((rok && c < cols) ? v.x : 0.f, (rok && c + 1 < cols) ? v.y : 0.f, (rok && c + 2 < cols) ? v.z : 0.f, (rok && c + 3 < cols) ? v.w : 0.f);
}

// one layer forward: nxt[16][N] = act(cur[16][K] . W[N, K]^T + b), also to Y
template <bool VEC>
__device__ __forceinline__ void tw_fwd_layer(const TowerArgs& a, int l, const float* cur, float* nxt, float* wst, int m0, int lane, int wave) {
    const int li = lane & 15, g = lane >> 4, tid = threadIdx.x;
    const int K = a.width[l], N = a.width[l + 1];
    const int LD = a.ld_lds;
    const float* __restrict__ W = a.W[l];
    const long long ldw = a.ldw[l];
    const int pr = tid >> 3, pq = (tid & 7) * 4;              // panel element of this thread: rows pr + 128 i, k-quad pq
    for (int nb = 0; nb < N; nb += TW_PASS) {
        const int rows = (N - nb < TW_PASS) ? N - nb : TW_PASS;          // output columns of this pass
        const int nload = (rows + 127) >> 7;                               // 128-row slabs of the panel that hold anything
        const bool t1 = (wave + TW_NW) * 16 < rows;                        // this wave's second tile exists
        const bool t0 = wave * 16 < rows;
        floatx4 acc[TW_TPW];
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
        // panels travel global -> registers -> LDS, requested TWO chunks ahead (one chunk of products does not cover an L2 round trip):
        // while chunk c is multiplied, chunk c+1 waits in one register set and chunk c+2 is in flight into the other.  The body handles
        // two chunks so that the sets keep their names (no copies); requests past the end are clamped and never parked.
        float4 preA[4], preB[4];
        auto fetch = [&](int kc, float4 (&pre)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nload) pre[i] = tw_panel_quad<VEC>(W, ldw, nb + pr + 128 * i, kc + pq, N, K);        // (wave-uniform condition)
        };
        auto park = [&](const float4 (&pre)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nload) *(float4*)(wst + (pr + 128 * i) * TW_WLD_F + pq) = pre[i];
        };
        auto products = [&](int kc) {
            if (!t0) return;
#pragma unroll
            for (int u = 0; u < TW_CK / 16; ++u) {
                const float4 a4 = *(const float4*)(cur + li * LD + kc + 16 * u + 4 * g);
                const float4 b0 = *(const float4*)(wst + (wave * 16 + li) * TW_WLD_F + 16 * u + 4 * g);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc[0], 0, 0, 0);
                if (t1) {
                    const float4 b1 = *(const float4*)(wst + ((wave + TW_NW) * 16 + li) * TW_WLD_F + 16 * u + 4 * g);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc[1], 0, 0, 0);
                }
            }
        };
        fetch(0, preA);
        fetch(TW_CK, preB);
        park(preA);
        __syncthreads();
        for (int kc = 0; kc < K; kc += 2 * TW_CK) {
            fetch(kc + 2 * TW_CK, preA);                                   // chunk c+2
            products(kc);                                                  // chunk c
            __syncthreads();                                               // everyone is done with this panel
            park(preB);                                                    // chunk c+1 (requested one chunk of products ago)
            __syncthreads();
            if (kc + TW_CK >= K) break;
            fetch(kc + 3 * TW_CK, preB);                                   // chunk c+3
            products(kc + TW_CK);                                          // chunk c+1
            __syncthreads();
            park(preA);                                                    // chunk c+2
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < TW_TPW; ++j) {
            if (!(j ? t1 : t0)) continue;
            const int n = nb + (wave + TW_NW * j) * 16 + li;
            const float b = (n < N && a.bias[l]) ? a.bias[l][n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * g + r;
                const float v = n < N ? tw_act(acc[j][r] + b, a.act[l]) : 0.f;
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
    float* wst = lds + 2 * TW_ROWS * LD;
    for (int e = threadIdx.x; e < 2 * TW_ROWS * LD; e += TW_THREADS) lds[e] = 0.f;       // finite everywhere (see above)
    __syncthreads();
    {   // the tower's input rows (zero beyond the batch)
        const int K0 = a.width[0];
        for (int e = threadIdx.x; e < TW_ROWS * K0; e += TW_THREADS) {
            const int r = e / K0, c = e - r * K0;
            if (m0 + r < a.M) cur[r * LD + c] = a.X[(long long)(m0 + r) * a.ldx + c];
        }
    }
    __syncthreads();
    for (int l = 0; l < a.L; ++l) {
        if (tw_vec_ok(a.W[l], a.ldw[l], a.width[l])) tw_fwd_layer<true>(a, l, cur, nxt, wst, m0, lane, wave);
        else tw_fwd_layer<false>(a, l, cur, nxt, wst, m0, lane, wave);
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }
}

// one layer backward: dA[16][K] = cur[16][N] (= dZ_l) . W[N, K]; then dZ_{l-1} = dA * act'_{l-1}(Y_{l-1}) -> nxt and dZ[l-1]   (l > 0)
//                                                                 or dX = dA                                              (l == 0)
// The reduction runs over ROWS of W: a chunk's panel is W[nc .. nc+31][0 .. K) as it lies in memory; MFMA s of a 16-wide step
// multiplies the reduction indices nc + 16u + 4g + s, its B operand B[k = g][j = li] = panel[16u + 4g + s][c + li] is a 4-byte LDS read.
template <bool VEC>
__device__ __forceinline__ void tw_bwd_layer(const TowerArgs& a, int l, const float* cur, float* nxt, float* wst, int m0, int lane, int wave) {
    const int li = lane & 15, g = lane >> 4, tid = threadIdx.x;
    const int K = a.width[l], N = a.width[l + 1];
    const int LD = a.ld_lds;
    const float* __restrict__ W = a.W[l];
    const long long ldw = a.ldw[l];
    const int ntiles = (K + 15) >> 4;
    const bool t0 = wave < ntiles, t1 = wave + TW_NW < ntiles;
    const int c0 = wave * 16 + li, c1 = (wave + TW_NW) * 16 + li;
    const int pq = (tid & 127) * 4;                           // panel element of this thread: rows (tid >> 7) + 8 i, column quad pq
    const int pr = tid >> 7;
    floatx4 acc[TW_TPW];
#pragma unroll
    for (int j = 0; j < TW_TPW; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float4 preA[4], preB[4];                                  // (two chunks ahead: see the forward loop)
    auto fetch = [&](int nc, float4 (&pre)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = tw_panel_quad<VEC>(W, ldw, nc + pr + 8 * i, pq, N, K);
    };
    auto park = [&](const float4 (&pre)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(float4*)(wst + (pr + 8 * i) * TW_WLD_B + pq) = pre[i];
    };
    auto products = [&](int nc) {
        if (!t0) return;
#pragma unroll
        for (int u = 0; u < TW_CK / 16; ++u) {
            const float4 a4 = *(const float4*)(cur + li * LD + nc + 16 * u + 4 * g);
            const float as[4] = {a4.x, a4.y, a4.z, a4.w};
            const float* prow = wst + (16 * u + 4 * g) * TW_WLD_B;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s_], prow[s_ * TW_WLD_B + c0], acc[0], 0, 0, 0);
            if (t1) {
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s_], prow[s_ * TW_WLD_B + c1], acc[1], 0, 0, 0);
            }
        }
    };
    fetch(0, preA);
    fetch(TW_CK, preB);
    park(preA);
    __syncthreads();
    for (int nc = 0; nc < N; nc += 2 * TW_CK) {
        fetch(nc + 2 * TW_CK, preA);
        products(nc);
        __syncthreads();
        park(preB);
        __syncthreads();
        if (nc + TW_CK >= N) break;
        fetch(nc + 3 * TW_CK, preB);
        products(nc + TW_CK);
        __syncthreads();
        park(preA);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TW_TPW; ++j) {
        if (!(j ? t1 : t0)) continue;
        const int c = j ? c1 : c0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float v = acc[j][r];
            if (l > 0) {
                const bool in = c < K && m0 + row < a.M;
                v = in ? tw_act_grad(v, a.Y[l - 1][(long long)(m0 + row) * a.ldy[l - 1] + c], a.act[l - 1]) : 0.f;
                nxt[row * LD + c] = v;
                if (in) a.dZ[l - 1][(long long)(m0 + row) * a.lddz[l - 1] + c] = v;
            } else if (c < K && m0 + row < a.M) {
                a.dX[(long long)(m0 + row) * a.lddx + c] = v;
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
    float* wst = lds + 2 * TW_ROWS * LD;
    for (int e = threadIdx.x; e < 2 * TW_ROWS * LD; e += TW_THREADS) lds[e] = 0.f;
    __syncthreads();
    {   // dZ of the last layer from the incoming gradient (a.X) and the layer's output
        const int l = a.L - 1, N = a.width[a.L];
        for (int e = threadIdx.x; e < TW_ROWS * N; e += TW_THREADS) {
            const int r = e / N, c = e - r * N;
            if (m0 + r < a.M) {
                float v = a.X[(long long)(m0 + r) * a.ldx + c];
                if (!a.last_act_applied) v = tw_act_grad(v, a.Y[l][(long long)(m0 + r) * a.ldy[l] + c], a.act[l]);
                a.dZ[l][(long long)(m0 + r) * a.lddz[l] + c] = v;
                cur[r * LD + c] = v;
            }
        }
    }
    __syncthreads();
    const int stop = a.dX ? 0 : 1;
    for (int l = a.L - 1; l >= stop; --l) {
        if (tw_vec_ok(a.W[l], a.ldw[l], a.width[l])) tw_bwd_layer<true>(a, l, cur, nxt, wst, m0, lane, wave);
        else tw_bwd_layer<false>(a, l, cur, nxt, wst, m0, lane, wave);
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
// invalidate the L2 of every XCD — 134 us for the launch, profiles/round5/step_trace_kaggle_graph_towers_v1.txt.)
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
    return ((w + 31) & ~31) + 4;          // whole 32-element chunks + 4: rows start 4 banks apart
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
    const size_t lds = ((size_t)2 * TW_ROWS * a.ld_lds + TW_WST_FLOATS) * sizeof(float);
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
    const size_t lds = ((size_t)2 * TW_ROWS * a.ld_lds + TW_WST_FLOATS) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)tower_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(tower_bwd_kernel, dim3((unsigned)((M + TW_ROWS - 1) / TW_ROWS)), dim3(TW_THREADS), lds, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// batch slices per tile: enough workgroups for two per CU, slices of at least 256 rows
static void tw_wgrad_plan(int64_t M, int L, const int* N, const int* K, int* tiles_out, int* S_out, int* rows_out) {
    int tiles = 0;
    for (int l = 0; l < L; ++l) tiles += ((N[l] + 63) / 64) * ((K[l] + 63) / 64);
    int S = (512 + tiles - 1) / tiles;
    const int64_t maxS = (M + 255) / 256;                   // slices of >= 256 rows: every slice is a 16.6 KB slab per tile to write and re-read
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
