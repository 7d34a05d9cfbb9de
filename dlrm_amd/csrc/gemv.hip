// gemv.hip — the N == 1 layer of an MLP tower (the last top-MLP layer of DLRM, 256 -> 1 + sigmoid,
// dlrm_s_pytorch.py:208-246): a matrix-vector product is pure HBM streaming, a 128-wide MFMA tile wastes 127/128 of the
// matrix pipe on it (measured with the GEMM kernels at M = 65536, K = 256: fwd 41 us, dgrad 43 us, wgrad 67 us for
// 67 MB of traffic each, i.e. 1.0-1.6 TB/s).
//   forward  y[m]     = act(<X[m,:], w> + b)                 one LPR-lane group per row, wave-shuffle reduction
//   dgrad    dX[m,k]  = dy[m] * w[k] * act'(Xact[m,k])       elementwise, 16 bytes per lane
//   wgrad    dW[k]    = sum_m dy[m] * X[m,k],  db = sum_m dy[m]   per-workgroup column partials + fixed-order finish
// Handled when K % 4 == 0, 16-byte aligned rows and K <= 1024; everything else falls through to the GEMM kernels.
#include "common.h"

namespace {

__device__ __forceinline__ float gv_act(float v, int act) {
    if (act == DLRM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DLRM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
__device__ __forceinline__ float gv_act_grad(float g, float y, int act) {
    if (act == DLRM_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DLRM_ACT_SIGMOID) return g * ((1.f - y) * y);
    return g;
}

constexpr int kNJ = 4;      // column chunks per lane: K <= 4 * kNJ * LPR
constexpr int kU = 4;       // rows in flight per lane group

template <int LPR>
__global__ __launch_bounds__(256) void gemv_fwd_kernel(long long M, int K, const float* __restrict__ X, long long ldx,
                                                       const float* __restrict__ w, const float* __restrict__ bias, int act,
                                                       float* __restrict__ Y, long long ldy) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, lig = lane % LPR;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (long long)gridDim.x * 4;
    const int kq = K >> 2;
    float4 wv[kNJ];
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
        const int c = j * LPR + lig;
        wv[j] = c < kq ? *(const float4*)(w + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float b = bias ? bias[0] : 0.f;
    for (long long r0 = wave * (RPW * kU); r0 < M; r0 += nwaves * (RPW * kU)) {
        float4 x[kU][kNJ];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const long long m = r0 + u * RPW + sub;
#pragma unroll
            for (int j = 0; j < kNJ; ++j) {
                const int c = j * LPR + lig;
                x[u][j] = (m < M && c < kq) ? *(const float4*)(X + m * ldx + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
                s += (x[u][j].x * wv[j].x + x[u][j].y * wv[j].y) + (x[u][j].z * wv[j].z + x[u][j].w * wv[j].w);
#pragma unroll
            for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const long long m = r0 + u * RPW + sub;
            if (lig == 0 && m < M) Y[m * ldy] = gv_act(s + b, act);
        }
    }
}

__global__ __launch_bounds__(256) void gemv_bwd_data_kernel(long long M, int K, const float* __restrict__ dY, long long lddy,
                                                            const float* __restrict__ w, const float* __restrict__ Xact,
                                                            long long ldxa, int kind, float* __restrict__ dX, long long lddx) {
    const int kq = K >> 2;
    const long long total = M * kq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / kq;
        const int c = (int)(e - m * kq);
        const float g = dY[m * lddy];
        const float4 wv = *(const float4*)(w + 4 * c);
        float4 v = make_float4(g * wv.x, g * wv.y, g * wv.z, g * wv.w);
        if (Xact) {
            const float4 y = *(const float4*)(Xact + m * ldxa + 4 * c);
            v.x = gv_act_grad(v.x, y.x, kind); v.y = gv_act_grad(v.y, y.y, kind);
            v.z = gv_act_grad(v.z, y.z, kind); v.w = gv_act_grad(v.w, y.w, kind);
        }
        *(float4*)(dX + m * lddx + 4 * c) = v;
    }
}

// stage 1: workgroup b sums its rows' dy[m] * X[m, :] into part[b][0..K) and dy[m] into part[b][K]
__global__ __launch_bounds__(256) void gemv_bwd_weight_partial_kernel(long long M, int K, const float* __restrict__ dY,
                                                                      long long lddy, const float* __restrict__ X,
                                                                      long long ldx, long long rows_per_block,
                                                                      float* __restrict__ part, int ldp) {
    __shared__ float4 red[256];
    __shared__ float redb[256];
    const int kq = K >> 2;                          // <= 256
    const int rg = 256 / kq;                        // row groups per pass (>= 1)
    const int cq = threadIdx.x % kq, g = threadIdx.x / kq;
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accb = 0.f;
    if (g < rg) {
        constexpr int UR = 8;                       // rows in flight per thread: the loop is latency-bound otherwise
        long long m = m0 + g;
        for (; m + (long long)(UR - 1) * rg < m1; m += (long long)UR * rg) {
            float d[UR];
            float4 x[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                d[u] = dY[(m + (long long)u * rg) * lddy];
                x[u] = *(const float4*)(X + (m + (long long)u * rg) * ldx + 4 * cq);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                acc.x = __builtin_fmaf(d[u], x[u].x, acc.x); acc.y = __builtin_fmaf(d[u], x[u].y, acc.y);
                acc.z = __builtin_fmaf(d[u], x[u].z, acc.z); acc.w = __builtin_fmaf(d[u], x[u].w, acc.w);
                accb += d[u];
            }
        }
        for (; m < m1; m += rg) {
            const float d = dY[m * lddy];
            const float4 x = *(const float4*)(X + m * ldx + 4 * cq);
            acc.x = __builtin_fmaf(d, x.x, acc.x); acc.y = __builtin_fmaf(d, x.y, acc.y);
            acc.z = __builtin_fmaf(d, x.z, acc.z); acc.w = __builtin_fmaf(d, x.w, acc.w);
            accb += d;
        }
    }
    red[threadIdx.x] = acc; redb[threadIdx.x] = accb;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < rg; ++q) {
            const float4 o = red[q * kq + cq];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            accb += redb[q * kq + cq];
        }
        *(float4*)(part + (long long)blockIdx.x * ldp + 4 * cq) = acc;
        if (cq == 0) part[(long long)blockIdx.x * ldp + K] = accb;
    }
}

// The whole backward of the N == 1 layer in ONE pass over X (dlrm_linear_head_bwd, round 6): the step ran act_bwd (dz = dy * act'(y), a launch
// over 65536 floats), the weight-gradient partials (X read: 67 MB) and the data gradient (X read again for the previous layer's ReLU mask +
// dX written: 134 MB) as three launches + the finish, 48 us for what one pass moves in ~22.  Same row partition, same accumulation order and the
// same expressions as gemv_bwd_weight_partial_kernel / gemv_bwd_data_kernel / act_bwd_kernel: partials and dX are BIT-IDENTICAL to theirs.
__global__ __launch_bounds__(256) void gemv_bwd_fused_partial_kernel(long long M, int K, const float* __restrict__ dY, long long lddy,
                                                                     const float* __restrict__ Yout, long long ldy, int act_out,
                                                                     const float* __restrict__ X, long long ldx,
                                                                     const float* __restrict__ w, int xact_kind,
                                                                     float* __restrict__ dX, long long lddx,
                                                                     long long rows_per_block, float* __restrict__ part, int ldp) {
    __shared__ float4 red[256];
    __shared__ float redb[256];
    const int kq = K >> 2;                          // <= 256
    const int rg = 256 / kq;                        // row groups per pass (>= 1)
    const int cq = threadIdx.x % kq, g = threadIdx.x / kq;
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float accb = 0.f;
    if (g < rg) {
        const float4 wv = *(const float4*)(w + 4 * cq);
        constexpr int UR = 8;
        long long m = m0 + g;
        for (; m + (long long)(UR - 1) * rg < m1; m += (long long)UR * rg) {
            float d[UR];
            float4 x[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long long r = m + (long long)u * rg;
                d[u] = dY[r * lddy];
                if (Yout) d[u] = gv_act_grad(d[u], Yout[r * ldy], act_out);
                x[u] = *(const float4*)(X + r * ldx + 4 * cq);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                acc.x = __builtin_fmaf(d[u], x[u].x, acc.x); acc.y = __builtin_fmaf(d[u], x[u].y, acc.y);
                acc.z = __builtin_fmaf(d[u], x[u].z, acc.z); acc.w = __builtin_fmaf(d[u], x[u].w, acc.w);
                accb += d[u];
                if (dX) {
                    float4 v = make_float4(d[u] * wv.x, d[u] * wv.y, d[u] * wv.z, d[u] * wv.w);
                    if (xact_kind != DLRM_ACT_NONE) {
                        v.x = gv_act_grad(v.x, x[u].x, xact_kind); v.y = gv_act_grad(v.y, x[u].y, xact_kind);
                        v.z = gv_act_grad(v.z, x[u].z, xact_kind); v.w = gv_act_grad(v.w, x[u].w, xact_kind);
                    }
                    *(float4*)(dX + (m + (long long)u * rg) * lddx + 4 * cq) = v;
                }
            }
        }
        for (; m < m1; m += rg) {
            float d = dY[m * lddy];
            if (Yout) d = gv_act_grad(d, Yout[m * ldy], act_out);
            const float4 x = *(const float4*)(X + m * ldx + 4 * cq);
            acc.x = __builtin_fmaf(d, x.x, acc.x); acc.y = __builtin_fmaf(d, x.y, acc.y);
            acc.z = __builtin_fmaf(d, x.z, acc.z); acc.w = __builtin_fmaf(d, x.w, acc.w);
            accb += d;
            if (dX) {
                float4 v = make_float4(d * wv.x, d * wv.y, d * wv.z, d * wv.w);
                if (xact_kind != DLRM_ACT_NONE) {
                    v.x = gv_act_grad(v.x, x.x, xact_kind); v.y = gv_act_grad(v.y, x.y, xact_kind);
                    v.z = gv_act_grad(v.z, x.z, xact_kind); v.w = gv_act_grad(v.w, x.w, xact_kind);
                }
                *(float4*)(dX + m * lddx + 4 * cq) = v;
            }
        }
    }
    red[threadIdx.x] = acc; redb[threadIdx.x] = accb;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < rg; ++q) {
            const float4 o = red[q * kq + cq];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            accb += redb[q * kq + cq];
        }
        *(float4*)(part + (long long)blockIdx.x * ldp + 4 * cq) = acc;
        if (cq == 0) part[(long long)blockIdx.x * ldp + K] = accb;
    }
}

// stage 2: fixed-order sum of the per-workgroup partials (deterministic): a workgroup owns 64 columns, its four
// thread rows each sum every fourth partial (8 loads in flight), LDS folds the four sums in a fixed order
__global__ __launch_bounds__(256) void gemv_bwd_weight_finish_kernel(int K, int nblk, const float* __restrict__ part, int ldp,
                                                                     float* __restrict__ dW, float* __restrict__ dbias,
                                                                     int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c <= K) {
        int b = q;
        for (; b + 28 < nblk; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(b + 4 * u) * ldp + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblk; b += 4) s += part[(long long)b * ldp + c];
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && c <= K) {
        s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (c < K) dW[c] = accumulate ? dW[c] + s : s;
        else if (dbias) dbias[0] = accumulate ? dbias[0] + s : s;
    }
}

static bool gemv_ok(const float* X, long long ldx, int K) {
    return K % 4 == 0 && K <= 1024 && dlrm_aligned16(X) && ldx % 4 == 0;
}

static void gemv_wgrad_plan(long long M, int K, int* nblk, long long* rows_per_block, int* ldp) {
    long long nb = (M + 127) / 128;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    *rows_per_block = (M + nb - 1) / nb;
    *nblk = (int)((M + *rows_per_block - 1) / *rows_per_block);
    *ldp = ((K + 1 + 3) & ~3);
}

}  // namespace

int64_t dlrm_gemv_bwd_weight_workspace_bytes(int64_t M, int K) {
    if (M <= 0 || K <= 0 || K % 4 != 0 || K > 1024) return 0;
    int nblk, ldp; long long rpb;
    gemv_wgrad_plan(M, K, &nblk, &rpb, &ldp);
    return (int64_t)nblk * ldp * (int64_t)sizeof(float);
}

// each returns 0 when it handled the call, DLRM_GEMV_NOT_HANDLED when the caller must use the GEMM path
int dlrm_gemv_fwd(int64_t M, int K, const float* X, int64_t ldx, const float* w, const float* bias, int act, float* Y,
                  int64_t ldy, hipStream_t st) {
    if (!gemv_ok(X, ldx, K) || !dlrm_aligned16(w)) return DLRM_GEMV_NOT_HANDLED;
    const int kq = K / 4;
    int lpr = 4; while (lpr < 64 && lpr < kq) lpr <<= 1;      // one 16-byte chunk per lane when the row fits a wave
    if (lpr * kNJ < kq) return DLRM_GEMV_NOT_HANDLED;
    const long long rows_per_wave_iter = (64 / lpr) * kU;
    long long blocks = (M + rows_per_wave_iter * 4 - 1) / (rows_per_wave_iter * 4);
    if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(256);
    switch (lpr) {
        case 4:  hipLaunchKernelGGL(gemv_fwd_kernel<4>, grid, block, 0, st, (long long)M, K, X, (long long)ldx, w, bias, act, Y, (long long)ldy); break;
        case 8:  hipLaunchKernelGGL(gemv_fwd_kernel<8>, grid, block, 0, st, (long long)M, K, X, (long long)ldx, w, bias, act, Y, (long long)ldy); break;
        case 16: hipLaunchKernelGGL(gemv_fwd_kernel<16>, grid, block, 0, st, (long long)M, K, X, (long long)ldx, w, bias, act, Y, (long long)ldy); break;
        case 32: hipLaunchKernelGGL(gemv_fwd_kernel<32>, grid, block, 0, st, (long long)M, K, X, (long long)ldx, w, bias, act, Y, (long long)ldy); break;
        default: hipLaunchKernelGGL(gemv_fwd_kernel<64>, grid, block, 0, st, (long long)M, K, X, (long long)ldx, w, bias, act, Y, (long long)ldy); break;
    }
    DLRM_LAUNCH_CHECK();
    return 0;
}

int dlrm_gemv_bwd_data(int64_t M, int K, const float* dY, int64_t lddy, const float* w, const float* Xact, int64_t ldxa,
                       int kind, float* dX, int64_t lddx, hipStream_t st) {
    if (!gemv_ok(dX, lddx, K) || !dlrm_aligned16(w)) return DLRM_GEMV_NOT_HANDLED;
    if (Xact && (!dlrm_aligned16(Xact) || ldxa % 4 != 0)) return DLRM_GEMV_NOT_HANDLED;
    const long long total = (long long)M * (K / 4);
    long long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gemv_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)M, K, dY, (long long)lddy, w,
                       Xact, (long long)ldxa, kind, dX, (long long)lddx);
    DLRM_LAUNCH_CHECK();
    return 0;
}

int dlrm_gemv_bwd_weight(int64_t M, int K, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW,
                         float* dbias, int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    if (!gemv_ok(X, ldx, K)) return DLRM_GEMV_NOT_HANDLED;
    const int64_t need = dlrm_gemv_bwd_weight_workspace_bytes(M, K);
    if (!workspace || !dlrm_aligned16(workspace) || workspace_bytes < need) return DLRM_GEMV_NOT_HANDLED;
    int nblk, ldp; long long rpb;
    gemv_wgrad_plan(M, K, &nblk, &rpb, &ldp);
    hipLaunchKernelGGL(gemv_bwd_weight_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (long long)M, K, dY,
                       (long long)lddy, X, (long long)ldx, rpb, (float*)workspace, ldp);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemv_bwd_weight_finish_kernel, dim3((unsigned)((K + 1 + 63) / 64)), dim3(256), 0, st, K, nblk,
                       (const float*)workspace, ldp, dW, dbias, accumulate ? 1 : 0);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// dz = dY * act'(Y) (Y == nullptr: dY is dz already), dW = dz^T X, db = sum dz, dX = (dz w) * xact'(X): one pass + the finish.
// DLRM_GEMV_NOT_HANDLED: the caller runs the three calls.
int dlrm_gemv_bwd_fused(int64_t M, int K, const float* dY, int64_t lddy, const float* Yout, int64_t ldy, int act_out, const float* X,
                        int64_t ldx, const float* w, int xact_kind, float* dX, int64_t lddx, float* dW, float* dbias, int accumulate,
                        void* workspace, int64_t workspace_bytes, hipStream_t st) {
    if (!gemv_ok(X, ldx, K) || !dlrm_aligned16(w) || K / 4 > 256) return DLRM_GEMV_NOT_HANDLED;
    if (dX && !gemv_ok(dX, lddx, K)) return DLRM_GEMV_NOT_HANDLED;
    const int64_t need = dlrm_gemv_bwd_weight_workspace_bytes(M, K);
    if (!workspace || !dlrm_aligned16(workspace) || workspace_bytes < need) return DLRM_GEMV_NOT_HANDLED;
    int nblk, ldp; long long rpb;
    gemv_wgrad_plan(M, K, &nblk, &rpb, &ldp);
    hipLaunchKernelGGL(gemv_bwd_fused_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (long long)M, K, dY, (long long)lddy, Yout,
                       (long long)ldy, act_out, X, (long long)ldx, w, xact_kind, dX, (long long)lddx, rpb, (float*)workspace, ldp);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemv_bwd_weight_finish_kernel, dim3((unsigned)((K + 1 + 63) / 64)), dim3(256), 0, st, K, nblk,
                       (const float*)workspace, ldp, dW, dbias, accumulate ? 1 : 0);
    DLRM_LAUNCH_CHECK();
    return 0;
}
