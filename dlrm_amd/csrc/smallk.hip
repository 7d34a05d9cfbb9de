// smallk.hip — weight gradient of MLP layers with a tiny input width (the first bottom-MLP layer of DLRM: 13 dense
// features, zero-padded to K = 16; dlrm_s_pytorch.py:208-246).  dW[N, 16] = dY^T X reduces over the whole batch into a
// 32 KB result: as a 256x128x16 MFMA GEMM it is one k-step per tile plus split-K slabs (measured at M = 65536, N = 512:
// 98 us for 134 MB read).  Here it streams:
//   dW[n,k] = sum_m dY[m,n] X[m,k], db[n] = sum_m dY[m,n]
//       a thread owns 4 consecutive n x K accumulators over a slab of rows (dY: one coalesced 16-byte load per row,
//       X row: scalar loads); slab partials + fixed-order finish (deterministic)                           -> 38 us
// The FORWARD of that layer stays on the GEMM kernel: a streaming version (thread = 4 outputs, weights in registers)
// measured 51 us against the GEMM's 42 us (it is bound by the 134 MB it writes, the MFMA tile hides the rest).
// Handled when K <= 16, K % 4 == 0, N % 4 == 0, N/4 divides 256 and rows are 16-byte aligned; else the GEMM path runs.
#include "common.h"

namespace {

// UNI: N/4 is a multiple of 64, so all lanes of a wave work on the same row: the row index is made wave-uniform
// (readfirstlane) and the X row arrives through scalar loads — no broadcast vector loads, no VGPRs for X
// stage 1: part[b] = [N][4*KQ] weight-gradient partial followed by [N] bias-gradient partial of the block's rows
template <int KQ, bool UNI>
__global__ __launch_bounds__(256) void smallk_wgrad_partial_kernel(long long M, int N, const float* __restrict__ dY,
                                                                   long long lddy, const float* __restrict__ X, long long ldx,
                                                                   long long rows_per_block, float* __restrict__ part,
                                                                   long long part_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [NQ][4*KQ*4 + 4] floats: one row group at a time
    constexpr int PER = 4 * KQ * 4 + 4;
    const int NQ = N >> 2, RPB = 256 / NQ;
    const int nq = threadIdx.x % NQ, rsub = threadIdx.x / NQ;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = (m_begin + rows_per_block < M) ? m_begin + rows_per_block : M;
    float4 acc[4][KQ];
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < KQ; ++k) acc[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;
    long long m = m_begin + rsub;
    auto add_row = [&](const float4& d, const float4 (&x)[KQ]) {
        const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                acc[j][k].x = __builtin_fmaf(dv[j], x[k].x, acc[j][k].x); acc[j][k].y = __builtin_fmaf(dv[j], x[k].y, acc[j][k].y);
                acc[j][k].z = __builtin_fmaf(dv[j], x[k].z, acc[j][k].z); acc[j][k].w = __builtin_fmaf(dv[j], x[k].w, acc[j][k].w);
            }
        accb.x += d.x; accb.y += d.y; accb.z += d.z; accb.w += d.w;
    };
    for (; m + (long long)(U - 1) * RPB < m_end; m += (long long)U * RPB) {
        float4 d[U], x[U][KQ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = m + (long long)u * RPB;
            const long long ru = UNI ? (long long)__builtin_amdgcn_readfirstlane((int)r) : r;
            d[u] = *(const float4*)(dY + r * lddy + 4 * nq);
#pragma unroll
            for (int k = 0; k < KQ; ++k) x[u][k] = *(const float4*)(X + ru * ldx + 4 * k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) add_row(d[u], x[u]);
    }
    for (; m < m_end; m += RPB) {
        float4 x[KQ];
        const float4 d = *(const float4*)(dY + m * lddy + 4 * nq);
#pragma unroll
        for (int k = 0; k < KQ; ++k) x[k] = *(const float4*)(X + m * ldx + 4 * k);
        add_row(d, x);
    }
    // fold the row groups in a fixed order (group 1, 2, ... into group 0), one group through LDS at a time
    for (int r = 1; r < RPB; ++r) {
        if (rsub == r) {
            float* p = lds + nq * PER;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < KQ; ++k) *(float4*)(p + (j * KQ + k) * 4) = acc[j][k];
            *(float4*)(p + 4 * KQ * 4) = accb;
        }
        __syncthreads();
        if (rsub == 0) {
            const float* p = lds + nq * PER;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const float4 o = *(const float4*)(p + (j * KQ + k) * 4);
                    acc[j][k].x += o.x; acc[j][k].y += o.y; acc[j][k].z += o.z; acc[j][k].w += o.w;
                }
            const float4 o = *(const float4*)(p + 4 * KQ * 4);
            accb.x += o.x; accb.y += o.y; accb.z += o.z; accb.w += o.w;
        }
        __syncthreads();
    }
    if (rsub == 0) {
        float* out = part + (long long)blockIdx.x * part_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KQ; ++k) *(float4*)(out + (long long)(4 * nq + j) * (4 * KQ) + 4 * k) = acc[j][k];
        *(float4*)(out + (long long)N * (4 * KQ) + 4 * nq) = accb;
    }
}

// stage 2: element e of [N*K | N]: fixed-order sum over the blocks' partials (4 interleaved sub-sums folded through LDS)
__global__ __launch_bounds__(256) void smallk_wgrad_finish_kernel(int N, int K, int K_store, int nblk, const float* __restrict__ part,
                                                                  long long part_stride, float* __restrict__ dW, long long lddw,
                                                                  float* __restrict__ dbias, int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long total = (long long)N * K + N;
    const long long e = (long long)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (e < total) {
        int b = q;
        for (; b + 28 < nblk; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(b + 4 * u) * part_stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblk; b += 4) s += part[(long long)b * part_stride + e];
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && e < total) {
        s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (e < (long long)N * K) {
            if ((int)(e % K) < K_store) {                       // columns K_store..K-1 are alignment padding of X
                float* o = dW + (e / K) * lddw + (e % K);
                *o = accumulate ? *o + s : s;
            }
        } else if (dbias) {
            float* o = dbias + (e - (long long)N * K);
            *o = accumulate ? *o + s : s;
        }
    }
}

static bool smallk_shape_ok(int N, int K) {
    if (K <= 0 || K > 16 || K % 4 != 0 || N < 4 || N % 4 != 0) return false;
    const int NQ = N / 4;
    return NQ <= 256 && 256 % NQ == 0;
}

static void smallk_wgrad_plan(long long M, int N, int K, int* nblk, long long* rows_per_block, long long* stride) {
    long long nb = (M + 255) / 256;                 // every partial is N*(K+1) floats: few, fat slabs
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    *rows_per_block = (M + nb - 1) / nb;
    *nblk = (int)((M + *rows_per_block - 1) / *rows_per_block);
    *stride = (long long)N * K + N;
}

}  // namespace

int64_t dlrm_smallk_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || !smallk_shape_ok(N, K)) return 0;
    int nblk; long long rpb, stride;
    smallk_wgrad_plan(M, N, K, &nblk, &rpb, &stride);
    return (int64_t)nblk * stride * (int64_t)sizeof(float);
}

int dlrm_smallk_bwd_weight(int64_t M, int N, int K, int K_store, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW,
                           int64_t lddw, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                           hipStream_t st) {
    if (!smallk_shape_ok(N, K) || !dlrm_aligned16(X) || !dlrm_aligned16(dY) || ldx % 4 || lddy % 4) return DLRM_GEMV_NOT_HANDLED;
    const int64_t need = dlrm_smallk_bwd_weight_workspace_bytes(M, N, K);
    if (!workspace || !dlrm_aligned16(workspace) || workspace_bytes < need) return DLRM_GEMV_NOT_HANDLED;
    int nblk; long long rpb, stride;
    smallk_wgrad_plan(M, N, K, &nblk, &rpb, &stride);
    const int KQ = K / 4, NQ = N / 4;
    const size_t lds = (size_t)NQ * (4 * KQ * 4 + 4) * sizeof(float);        // <= 64 x 68 x 4 ... 128 x 68 x 4 = 34.8 KB
    dim3 grid((unsigned)nblk), block(256);
    const bool uni = NQ % 64 == 0 && M < ((int64_t)1 << 31);
#define SKW(Q) (uni ? smallk_wgrad_partial_kernel<Q, true> : smallk_wgrad_partial_kernel<Q, false>)
    switch (KQ) {
        case 1: hipLaunchKernelGGL(SKW(1), grid, block, lds, st, (long long)M, N, dY, (long long)lddy, X, (long long)ldx, rpb, (float*)workspace, stride); break;
        case 2: hipLaunchKernelGGL(SKW(2), grid, block, lds, st, (long long)M, N, dY, (long long)lddy, X, (long long)ldx, rpb, (float*)workspace, stride); break;
        case 3: hipLaunchKernelGGL(SKW(3), grid, block, lds, st, (long long)M, N, dY, (long long)lddy, X, (long long)ldx, rpb, (float*)workspace, stride); break;
        default: hipLaunchKernelGGL(SKW(4), grid, block, lds, st, (long long)M, N, dY, (long long)lddy, X, (long long)ldx, rpb, (float*)workspace, stride); break;
    }
    DLRM_LAUNCH_CHECK();
    const long long total = (long long)N * K + N;
    hipLaunchKernelGGL(smallk_wgrad_finish_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, N, K, K_store, nblk,
                       (const float*)workspace, stride, dW, (long long)lddw, dbias, accumulate ? 1 : 0);
    DLRM_LAUNCH_CHECK();
    return 0;
}
