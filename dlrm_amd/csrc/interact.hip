// interact.hip — fused pairwise-dot feature interaction, forward and backward, gfx950.
//
// Reference call site replaced: DLRM_Net.interact_features, arch_interaction_op == "dot"
//   (dlrm_s_pytorch.py:483-504): cat -> bmm(T, T^T) -> Z[:, li, lj] -> cat.
//
// One wavefront owns one sample at a time.  The sample's F feature vectors (F = 1 + #tables,
// each D floats, addressed through a per-feature {pointer, stride} table so the same kernel reads
// the single-GPU [B, F, D] buffer or the all-to-all receive buffer in place) are staged once in
// the wave's private LDS region, zero-padded to 16-row / 16-column multiples.
//
// forward : Z = T·T^T on v_mfma_f32_16x16x4_f32, only the lower-triangular 16x16 tile pairs
//           (3 of 4 for F = 27), A and B fragments both read as one ds_read_b128 per 16 k-values
//           (the k order inside an MFMA chain is permuted identically for A and B, which is
//           legal because both come from the same LDS image); emits R = [x | tril(Z)] directly.
// backward: S = dZ + dZ^T is rebuilt in LDS from dR, dT = S·T on the same MFMA; dT rows are
//           written through a second {pointer, stride} table (bottom-MLP grad buffer, embedding
//           grad / all-to-all send buffer), with dR[:, 0:D] added into row 0.
#include "common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
// pointers that round-trip through LDS as integers lose their address space; tag them global so
// the loads/stores are global_* (not flat_*) instructions
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) floatx4 gfloatx4;

#define DLRM_MAX_FEATURES 64
struct FeatArgs {
    const float* p[DLRM_MAX_FEATURES];
    long long    ld[DLRM_MAX_FEATURES];
};

// copy the kernarg pointer table into LDS with compile-time kernarg offsets (a lane-indexed read
// of a by-value struct would otherwise be demoted to scratch memory)
__device__ __forceinline__ void table_to_lds(const FeatArgs& fa, long long* tp, long long* tl, int F) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int f = 0; f < DLRM_MAX_FEATURES; ++f) {
        if (tid == f && f < F) { tp[f] = (long long)fa.p[f]; tl[f] = fa.ld[f]; }
    }
}

// stage one sample's features into the wave's LDS region (row stride LS floats)
__device__ __forceinline__ void stage_sample(float* my, const long long* tp, const long long* tl,
                                             long long b, int F, int D, int LS, int lane, int vec,
                                             int d4shift) {
    if (vec) {
        const int D4 = D >> 2;
        const int n = F * D4;
        for (int e = lane; e < n; e += 64) {
            int f, c;
            if (d4shift >= 0) { f = e >> d4shift; c = e & (D4 - 1); } else { f = e / D4; c = e - f * D4; }
            const floatx4 v = *(const gfloatx4*)((const gfloat*)tp[f] + b * tl[f] + 4 * c);
            *(floatx4*)__builtin_assume_aligned(my + f * LS + 4 * c, 16) = v;
        }
    } else {
        const int n = F * D;
        for (int e = lane; e < n; e += 64) {
            const int f = e / D, c = e - f * D;
            my[f * LS + c] = ((const gfloat*)tp[f])[b * tl[f] + c];
        }
    }
}

// -------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interact_fwd_kernel(FeatArgs fa, long long B, int F, int D, int self,
                                                           float* __restrict__ R, long long ldr, int vec,
                                                           int d4shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Dp = (D + 15) & ~15;
    const int LS = Dp + 4;            // 16-byte aligned rows, b128 reads of 16 rows hit 16 slots
    const int NB = (F + 15) >> 4;
    long long* tp = (long long*)lds;
    long long* tl = tp + DLRM_MAX_FEATURES;
    float* my = (float*)(tl + DLRM_MAX_FEATURES) + (size_t)wave * (NB * 16) * LS;

    table_to_lds(fa, tp, tl, F);
    for (int e = lane; e < NB * 16 * LS; e += 64) my[e] = 0.f;   // padding rows / columns stay zero
    __syncthreads();

    const int g = lane >> 4, li = lane & 15;
    const int P = self ? F * (F + 1) / 2 : F * (F - 1) / 2;
    const int nsteps = Dp >> 4;

    for (long long base = (long long)blockIdx.x * 4; base < B; base += (long long)gridDim.x * 4) {
        const long long b = base + wave;
        const bool valid = b < B;
        if (valid) stage_sample(my, tp, tl, b, F, D, LS, lane, vec, d4shift);
        __syncthreads();
        if (valid) {
            float* Rb = R + b * ldr;
            for (int r = 0; r < NB; ++r) {
                for (int c = 0; c <= r; ++c) {
                    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* ap = my + (16 * r + li) * LS + 4 * g;
                    const float* bp = my + (16 * c + li) * LS + 4 * g;
                    for (int s = 0; s < nsteps; ++s) {
                        const float4 av = *(const float4*)__builtin_assume_aligned(ap + 16 * s, 16);
                        const float4 bv = *(const float4*)__builtin_assume_aligned(bp + 16 * s, 16);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
                    }
                    const floatx4 acc = acc0 + acc1;
                    const int j = 16 * c + li;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 16 * r + 4 * g + q;   // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
                        if (i < F && (self ? (j <= i) : (j < i))) {
                            const int p = (self ? i * (i + 1) / 2 : i * (i - 1) / 2) + j;
                            Rb[D + p] = acc[q];
                        }
                    }
                }
            }
            for (int d = lane; d < D; d += 64) Rb[d] = my[d];            // R[:, 0:D] = x
            for (long long d = D + P + lane; d < ldr; d += 64) Rb[d] = 0.f;  // alignment padding
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void interact_bwd_kernel(FeatArgs fa, FeatArgs da, long long B, int F, int D,
                                                           int self, const float* __restrict__ dR,
                                                           long long ldr, int vec, int d4shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Dp = (D + 15) & ~15;
    const int LS = Dp + 16;           // row stride % 32 == 16: b32 column reads of 2 rows per half-wave are conflict free
    constexpr int rows = NB * 16;
    constexpr int SS = rows + 1;      // S row stride
    long long* tp = (long long*)lds;
    long long* tl = tp + DLRM_MAX_FEATURES;
    long long* dp = tl + DLRM_MAX_FEATURES;
    long long* dl = dp + DLRM_MAX_FEATURES;
    float* wbase = (float*)(dl + DLRM_MAX_FEATURES) + (size_t)wave * (F * LS + F * SS);
    float* my = wbase;                // staged features, F rows (rows >= F read as 0)
    float* S = wbase + F * LS;        // symmetric gradient matrix, F rows x `rows` columns

    table_to_lds(fa, tp, tl, F);
    table_to_lds(da, dp, dl, F);
    for (int e = lane; e < F * LS; e += 64) my[e] = 0.f;
    __syncthreads();

    const int g = lane >> 4, li = lane & 15;
    const int nct = Dp >> 4;

    for (long long base = (long long)blockIdx.x * 4; base < B; base += (long long)gridDim.x * 4) {
        const long long b = base + wave;
        const bool valid = b < B;
        if (valid) {
            stage_sample(my, tp, tl, b, F, D, LS, lane, vec, d4shift);
            const float* dRb = dR + b * ldr;
            for (int e = lane; e < F * rows; e += 64) {
                const int i = e / rows, j = e - i * rows;
                float v = 0.f;
                if (j < F) {
                    if (i == j) {
                        if (self) v = 2.f * dRb[D + i * (i + 1) / 2 + i];
                    } else {
                        const int hi = i > j ? i : j, lo = i > j ? j : i;
                        v = dRb[D + (self ? hi * (hi + 1) / 2 : hi * (hi - 1) / 2) + lo];
                    }
                }
                S[i * SS + j] = v;
            }
        }
        __syncthreads();
        if (valid) {
            // A fragments: S[16r + li][4kk + g]
            float aS[NB][4 * NB];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int kk = 0; kk < 4 * NB; ++kk)
                {
                    const int ir = 16 * r + li;
                    const float sv = S[(ir < F ? ir : 0) * SS + 4 * kk + g];   // unconditional load, then select
                    aS[r][kk] = (ir < F) ? sv : 0.f;
                }
            // destination row pointers for this lane's 4*NB output rows
            gfloat* orow[NB][4];
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * r + 4 * g + q;
                    orow[r][q] = (i < F) ? ((gfloat*)dp[i] + b * dl[i]) : nullptr;
                }
            const float* dRb = dR + b * ldr;
            for (int dc = 0; dc < nct; ++dc) {
                float bT[4 * NB];
#pragma unroll
                for (int kk = 0; kk < 4 * NB; ++kk)
                {
                    const int jr = 4 * kk + g;
                    const float tv = my[(jr < F ? jr : 0) * LS + 16 * dc + li];
                    bT[kk] = (jr < F) ? tv : 0.f;
                }
                const int d = 16 * dc + li;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 4 * NB; kk += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk], bT[kk], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk + 1], bT[kk + 1], acc1, 0, 0, 0);
                    }
                    const floatx4 acc = acc0 + acc1;
                    if (d < D) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (orow[r][q]) {
                                float v = acc[q];
                                if (r == 0 && q == 0 && g == 0) v += dRb[d];   // feature 0 also feeds R[:, 0:D]
                                orow[r][q][d] = v;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

static int fill_feat(FeatArgs& fa, int F, const void* const* p, const int64_t* ld) {
    for (int f = 0; f < DLRM_MAX_FEATURES; ++f) {
        fa.p[f] = (const float*)p[f < F ? f : 0];
        fa.ld[f] = ld[f < F ? f : 0];
        if (f < F && !p[f]) return DLRM_E_ARG;
    }
    return 0;
}

static int log2_exact(int x) {
    if (x <= 0 || (x & (x - 1))) return -1;
    int s = 0; while ((1 << s) < x) ++s; return s;
}

static int pick_grid(int64_t B) {
    // 4 samples per workgroup pass; ~2 resident workgroups per CU x 256 CUs x 4 oversubscription
    int64_t nb = (B + 3) / 4;
    if (nb > 2048) nb = 2048;
    return (int)nb;
}

}  // namespace

extern "C" int dlrm_interact_fwd(int64_t B, int F, int D, const void* const* feat_host,
                                 const int64_t* feat_ld_host, int self_interaction, float* R,
                                 int64_t ldr, void* stream) {
    if (B <= 0 || F <= 0 || D <= 0 || !feat_host || !feat_ld_host || !R) return DLRM_E_ARG;
    if (F > DLRM_MAX_FEATURES) {
        fprintf(stderr, "libdlrm_hip: dlrm_interact_fwd: F=%d exceeds %d features\n", F, DLRM_MAX_FEATURES);
        return DLRM_E_RANGE;
    }
    const int P = self_interaction ? F * (F + 1) / 2 : F * (F - 1) / 2;
    if (ldr < D + P) return DLRM_E_ARG;
    FeatArgs fa;
    int rc = fill_feat(fa, F, feat_host, feat_ld_host);
    if (rc) return rc;
    int vec = (D % 4 == 0);
    for (int f = 0; f < F; ++f) vec = vec && dlrm_aligned16(feat_host[f]) && (feat_ld_host[f] % 4 == 0);
    const int Dp = (D + 15) & ~15;
    const size_t lds = 2 * DLRM_MAX_FEATURES * sizeof(long long) + 4 * (size_t)(((F + 15) >> 4) * 16) * (Dp + 4) * sizeof(float);
    if (lds > 160 * 1024) {
        fprintf(stderr, "libdlrm_hip: dlrm_interact_fwd: F=%d, D=%d needs %zu B of LDS (> 160 KiB)\n", F, D, lds);
        return DLRM_E_RANGE;
    }
    (void)hipFuncSetAttribute((const void*)interact_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(interact_fwd_kernel, dim3(pick_grid(B)), dim3(256), lds, (hipStream_t)stream, fa,
                       (long long)B, F, D, self_interaction ? 1 : 0, R, (long long)ldr, vec,
                       vec ? log2_exact(D / 4) : -1);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_interact_bwd(int64_t B, int F, int D, const void* const* feat_host,
                                 const int64_t* feat_ld_host, int self_interaction, const float* dR,
                                 int64_t ldr, void* const* dfeat_host, const int64_t* dfeat_ld_host,
                                 void* stream) {
    if (B <= 0 || F <= 0 || D <= 0 || !feat_host || !feat_ld_host || !dR || !dfeat_host || !dfeat_ld_host)
        return DLRM_E_ARG;
    if (F > DLRM_MAX_FEATURES) return DLRM_E_RANGE;
    const int P = self_interaction ? F * (F + 1) / 2 : F * (F - 1) / 2;
    if (ldr < D + P) return DLRM_E_ARG;
    FeatArgs fa, da;
    int rc = fill_feat(fa, F, feat_host, feat_ld_host);
    if (rc) return rc;
    rc = fill_feat(da, F, (const void* const*)dfeat_host, dfeat_ld_host);
    if (rc) return rc;
    int vec = (D % 4 == 0);
    for (int f = 0; f < F; ++f) vec = vec && dlrm_aligned16(feat_host[f]) && (feat_ld_host[f] % 4 == 0);
    const int Dp = (D + 15) & ~15, NB = (F + 15) >> 4, rows = NB * 16;
    const size_t lds = 4 * DLRM_MAX_FEATURES * sizeof(long long) +
                       4 * ((size_t)F * (Dp + 16) + (size_t)F * (rows + 1)) * sizeof(float);
    if (lds > 160 * 1024) {
        fprintf(stderr, "libdlrm_hip: dlrm_interact_bwd: F=%d, D=%d needs %zu B of LDS (> 160 KiB)\n", F, D, lds);
        return DLRM_E_RANGE;
    }
    const int d4s = vec ? log2_exact(D / 4) : -1;
    dim3 grid(pick_grid(B)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define BWD_LAUNCH(NBV)                                                                                  \
    do {                                                                                                 \
        (void)hipFuncSetAttribute((const void*)interact_bwd_kernel<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(interact_bwd_kernel<NBV>, grid, block, lds, st, fa, da, (long long)B, F, D,   \
                           self_interaction ? 1 : 0, dR, (long long)ldr, vec, d4s);                      \
    } while (0)
    switch (NB) {
        case 1: BWD_LAUNCH(1); break;
        case 2: BWD_LAUNCH(2); break;
        case 3: BWD_LAUNCH(3); break;
        case 4: BWD_LAUNCH(4); break;
        default: return DLRM_E_RANGE;
    }
#undef BWD_LAUNCH
    DLRM_LAUNCH_CHECK();
    return 0;
}
