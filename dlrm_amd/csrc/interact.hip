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
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
// pointers that round-trip through LDS as integers lose their address space; tag them global so
// the loads/stores are global_* (not flat_*) instructions
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) floatx4 gfloatx4;
typedef __attribute__((address_space(1))) char gchar;          // a pointer rebuilt from an integer table is GENERIC unless told otherwise: its stores
                                                               // would be FLAT instructions, which also count on lgkmcnt — every LDS wait then waits for them

// position of the pair (i, j), j <= i, in the flattened interaction output.  `self` is a mode word: bit 0 = pairs include the
// diagonal (--arch-interaction-itself), bit 1 = torchrec order — torch.triu_indices(F, F, offset=1), i.e. pair (j, i) of the
// upper triangle enumerated row by row (torchrec InteractionArch) — instead of the reference's tril order (dlrm_s_pytorch.py:499-501).
__device__ __forceinline__ int pair_pos(int i, int j, int F, int mode) {
    if (mode & 2) return j * F - j * (j + 1) / 2 + (i - j - 1);
    return ((mode & 1) ? i * (i + 1) / 2 : i * (i - 1) / 2) + j;
}

#define DLRM_MAX_FEATURES 64
struct FeatArgs {
    const float* p[DLRM_MAX_FEATURES];
    long long    ld[DLRM_MAX_FEATURES];
};

// "gather" mode of the D = 128 kernels: a feature whose idx[f] != NULL is NOT a [B, D] matrix but a one-hot EmbeddingBag —
// row b of the feature is table row idx[f][b] (p[f] = table base, ld[f] = D).  The pooled-embedding buffer between
// dlrm_emb_fwd and the interaction (a 852 MB write + 852 MB read at Criteo-Terabyte shapes) then never exists.
// Round 2's version was SLOWER than the two kernels (forward 0.64 ms vs 0.30 + 0.24, backward 0.87 vs 0.38): its row selectors came in
// through ordinary loads, whose first use drains the whole VM queue beside an LDS-DMA in flight.  They now arrive by LDS-DMA themselves,
// three samples ahead (see "gather mode" below).
struct GatherArgs {
    const void* idx[DLRM_MAX_FEATURES];     // NULL: plain feature matrix
    const void* off[DLRM_MAX_FEATURES];     // bag starts of that table: verified to be 0, 1, 2, ... (one lookup per bag)
    long long   rows[DLRM_MAX_FEATURES];
    long long*  err;
    int         idx_bits;
    DlrmPred    pred;                       // launch predicate of the *_pred entry points (both D = 128 kernels, gather or not, take this struct)
};

__device__ __forceinline__ void gather_to_lds(const GatherArgs& ga, long long* tq, long long* to, long long* tr, int F) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int f = 0; f < DLRM_MAX_FEATURES; ++f) {
        if (tid == f && f < F) { tq[f] = (long long)ga.idx[f]; to[f] = (long long)ga.off[f]; tr[f] = ga.rows[f]; }
    }
}

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
    const int P = (self & 1) ? F * (F + 1) / 2 : F * (F - 1) / 2;
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
                        if (i < F && ((self & 1) ? (j <= i) : (j < i))) {
                            const int p = pair_pos(i, j, F, self);
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
                        if (self & 1) v = 2.f * dRb[D + pair_pos(i, i, F, self)];
                    } else {
                        const int hi = i > j ? i : j, lo = i > j ? j : i;
                        v = dRb[D + pair_pos(hi, lo, F, self)];
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
                                if (r == 0 && q == 0 && g == 0) {
                                    v += dRb[d];                               // feature 0 also feeds R[:, 0:D]
                                    if ((self & 4) && !(my[d] > 0.f)) v = 0.f; // ... and is a ReLU output whose derivative is applied here
                                }
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

// =============================================================================================
// D = 128 fast path: features arrive by LDS-DMA, double buffered per wave.
//
// What limited the kernels above (profiles/r01): one wave stages its sample through registers
// (13.5 dependent load -> ds_write rounds), then computes, then stores: ~14 us per sample per wave, 2.3 TB/s.
// Here every wave owns two 16-KiB LDS images [32 rows][128 floats]; the F feature rows of the NEXT sample
// are fetched by ceil(F/2) `global_load_lds_dwordx4` (two 512-B rows per instruction, all in flight at once)
// while the MFMAs of the current sample run; a counted `s_waitcnt vmcnt` retires exactly the current
// sample.  Images are wave private: no barriers in the sample loop.  16-byte slot q of row r holds feature
// quad q ^ (r & 15) (swizzle applied to the per-lane SOURCE address, the DMA destination is lane-linear):
// the ds_read_b128 fragment reads of both kernels are conflict free without padding.  Rows F..31 are zeroed
// once and never written again (lanes of the odd last row are masked out of the DMA).
// =============================================================================================
constexpr int IDMA_D = 128;
constexpr int IDMA_ROWS = 32;
constexpr int IDMA_IMG = IDMA_ROWS * IDMA_D * 4;        // 16 KiB
constexpr int IDMA_MAXI = IDMA_ROWS / 2;                // DMA instructions per sample (two rows each)

__device__ __forceinline__ void glds16_v(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt_i() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wave-uniform runtime count (0..40; larger counts wait for more than asked: safe) -> immediate
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
    switch (n) {
        case 0: wait_vmcnt_i<0>(); break;   case 1: wait_vmcnt_i<1>(); break;   case 2: wait_vmcnt_i<2>(); break;
        case 3: wait_vmcnt_i<3>(); break;   case 4: wait_vmcnt_i<4>(); break;   case 5: wait_vmcnt_i<5>(); break;
        case 6: wait_vmcnt_i<6>(); break;   case 7: wait_vmcnt_i<7>(); break;   case 8: wait_vmcnt_i<8>(); break;
        case 9: wait_vmcnt_i<9>(); break;   case 10: wait_vmcnt_i<10>(); break; case 11: wait_vmcnt_i<11>(); break;
        case 12: wait_vmcnt_i<12>(); break; case 13: wait_vmcnt_i<13>(); break; case 14: wait_vmcnt_i<14>(); break;
        case 15: wait_vmcnt_i<15>(); break; case 16: wait_vmcnt_i<16>(); break; case 17: wait_vmcnt_i<17>(); break;
        case 18: wait_vmcnt_i<18>(); break; case 19: wait_vmcnt_i<19>(); break; case 20: wait_vmcnt_i<20>(); break;
        case 21: wait_vmcnt_i<21>(); break; case 22: wait_vmcnt_i<22>(); break; case 23: wait_vmcnt_i<23>(); break;
        case 24: wait_vmcnt_i<24>(); break; case 25: wait_vmcnt_i<25>(); break; case 26: wait_vmcnt_i<26>(); break;
        case 27: wait_vmcnt_i<27>(); break; case 28: wait_vmcnt_i<28>(); break; case 29: wait_vmcnt_i<29>(); break;
        case 30: wait_vmcnt_i<30>(); break; case 31: wait_vmcnt_i<31>(); break; case 32: wait_vmcnt_i<32>(); break;
        case 33: wait_vmcnt_i<33>(); break; case 34: wait_vmcnt_i<34>(); break; case 35: wait_vmcnt_i<35>(); break;
        case 36: wait_vmcnt_i<36>(); break; case 37: wait_vmcnt_i<37>(); break; case 38: wait_vmcnt_i<38>(); break;
        case 39: wait_vmcnt_i<39>(); break; default: wait_vmcnt_i<40>(); break;
    }
}

struct DmaPlan {            // per lane: source pointer of its 16 bytes in each of the sample's DMA instructions
    const char* src[IDMA_MAXI];
    long long step[IDMA_MAXI];   // bytes from one of this wave's samples to its next
    bool on[IDMA_MAXI];
};

__device__ __forceinline__ void dma_plan_init(DmaPlan& pl, const long long* tp, const long long* tl, int F, int lane,
                                              long long b_first, long long b_stride) {
#pragma unroll
    for (int c = 0; c < IDMA_MAXI; ++c) {
        const int row = 2 * c + (lane >> 5);
        const int quad = (lane & 31) ^ (row & 15);
        pl.on[c] = row < F;
        const int rr = pl.on[c] ? row : 0;
        pl.src[c] = (const char*)tp[rr] + (b_first * tl[rr] + 4 * quad) * 4;
        pl.step[c] = b_stride * tl[rr] * 4;
    }
}

template <int NI>
__device__ __forceinline__ void dma_issue(DmaPlan& pl, unsigned lds_img) {
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        if (pl.on[c]) glds16_v(pl.src[c], lds_img + c * 1024);
        pl.src[c] += pl.step[c];
    }
}

// ---- gather mode -------------------------------------------------------------------------------------------------------
// The row selector of feature f for sample b is b itself for a plain feature and idx[f][b] for a gathered one.  Selectors never
// travel through registers on their way in: an ordinary global_load beside an LDS-DMA in flight makes the compiler drain the
// whole VM queue at its first use (round 2's version did exactly that: 0.64 ms for the fused forward).  Lane l of ONE
// `global_load_lds_dword` fetches the low (l < 32) or high (l >= 32) dword of idx[l & 31][b] into a 256-byte LDS slot, a second
// one the bag start off[l & 31][b] (verified to equal b: one lookup per bag); both are issued three samples ahead, so by the
// time a sample's rows are addressed its selectors are older than everything the counted wait leaves in flight.
constexpr int GSEL_SLOT = 512;                          // idx dwords [64] | off dwords [64]
constexpr int GSEL_WAVE_BYTES = 3 * GSEL_SLOT;          // ring of three samples per wave
constexpr int GDR_BYTES = 2048;                         // dR row image in gather mode (dlrm_interact_gather_ok bounds the row)

__device__ __forceinline__ void glds4_v(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NI>
struct GatherCtx {
    const char* base[NI];        // per DMA chunk: byte address of selector 0 of this lane's row, incl. the lane's swizzled 16 bytes
    unsigned stride[NI];         // bytes per selector
    bool on[NI];
    const char* qsrc;            // this lane's dword of idx[lane & 31][0] / off[lane & 31][0]
    const char* osrc;
    long long rows;              // of feature lane & 31; < 0: plain feature (selector = the sample number) or no feature
    int esz;                     // bytes per index
    int oesz;                    // bytes per element of this lane's offsets stream (lane 63 may read a uint32 per sample instead: `words`)
};
constexpr int GSEL_WORD_OFF = 256 + 4 * 63;      // where lane 63's dword of the second selector load lands inside a slot

template <int NI>
__device__ __forceinline__ void gather_ctx_init(GatherCtx<NI>& gc, const long long* tp, const long long* tl, const long long* tq,
                                                const long long* to, const long long* tr, int F, int lane, int idx_bits,
                                                const unsigned* words = nullptr) {
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        const int row = 2 * c + (lane >> 5);
        const int quad = (lane & 31) ^ (row & 15);
        gc.on[c] = row < F;
        const int rr = gc.on[c] ? row : 0;
        gc.base[c] = (const char*)tp[rr] + 16 * quad;
        gc.stride[c] = (unsigned)(tl[rr] * 4);
    }
    const int f = lane & 31, ff = f < F ? f : 0;
    const int hi = (idx_bits == 64) ? 4 * (lane >> 5) : 0;
    gc.qsrc = (const char*)tq[ff] + hi;
    gc.osrc = (const char*)to[ff] + hi;
    gc.rows = f < F ? tr[f] : -1;
    gc.esz = idx_bits >> 3;
    gc.oesz = gc.esz;
    // one uint32 per sample rides in on the offsets load: lane 63 stands for feature 31, which no gather call has (F <= 27), so its dword of
    // that load is free — no extra VM operation, no extra LDS (the fused update's single_mask: interact_bwd_dma_kernel<.., UPD>)
    if (words && lane == 63) { gc.osrc = (const char*)words; gc.oesz = 4; }
}

// selectors of sample s (any s < B: the caller clamps look-ahead past the end) -> LDS slot; two VM operations
template <int NI>
__device__ __forceinline__ void gather_sel_issue(const GatherCtx<NI>& gc, long long s, unsigned slot_lds) {
    glds4_v(gc.qsrc + s * gc.esz, slot_lds);
    glds4_v(gc.osrc + s * gc.oesz, slot_lds + 256);
}

// rows of sample s from its (landed) selector slot -> image; NI VM operations.  Verifies the one-lookup-per-bag layout and the
// index range in the lane that owns the feature; a violation is reported and row 0 is read instead.
// (the four selector dwords of a lane: read from a LANDED slot one sample ahead of their use, so that a sample's row loads are issued from
// registers at the top of the iteration instead of behind an LDS round trip — one wave per SIMD, nothing else hides it)
struct GatherSel { unsigned lo, hi, olo, ohi; };
__device__ __forceinline__ GatherSel gather_sel_read(const char* slot, int lane) {
    const int f = lane & 31;
    GatherSel g;
    g.lo = *(const unsigned*)(slot + 4 * f); g.hi = *(const unsigned*)(slot + 128 + 4 * f);
    g.olo = *(const unsigned*)(slot + 256 + 4 * f); g.ohi = *(const unsigned*)(slot + 384 + 4 * f);
    return g;
}

template <int NI>
__device__ __forceinline__ void gather_rows_issue(const GatherCtx<NI>& gc, const GatherSel& gs, long long s, int lane, int idx_bits,
                                                  long long* err, unsigned img_lds) {
    const int f = lane & 31;
    const unsigned lo = gs.lo, hi = gs.hi, olo = gs.olo, ohi = gs.ohi;
    unsigned idu = (unsigned)s;
    if (gc.rows >= 0) {
        long long id = idx_bits == 64 ? (long long)(((unsigned long long)hi << 32) | lo) : (long long)(int)lo;
        const long long o = idx_bits == 64 ? (long long)(((unsigned long long)ohi << 32) | olo) : (long long)(int)olo;
        if (o != s) dlrm_report_bad_index(err, f - 1, -(o + 1), -1);                    // not a one-lookup-per-bag batch (rows = -1 marks it)
        if (!dlrm_index_ok(id, gc.rows)) { dlrm_report_bad_index(err, f - 1, id, gc.rows); id = 0; }
        idu = (unsigned)id;
    }
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        const unsigned a = __builtin_amdgcn_readlane(idu, 2 * c), b2 = __builtin_amdgcn_readlane(idu, (2 * c + 1) & 31);
        const unsigned mine = (lane >> 5) ? b2 : a;
        if (gc.on[c]) glds16_v(gc.base[c] + (unsigned long long)mine * gc.stride[c], img_lds + c * 1024);
    }
}

template <int NI, bool GATHER>       // NI = ceil(F / 2)
__global__ __launch_bounds__(320) void interact_fwd_dma_kernel(FeatArgs fa, GatherArgs ga, long long B, int F, int self,
                                                               float* __restrict__ R, long long ldr) {
    if (ga.pred.skip()) return;                              // (*_pred entry points: the other implementation of this step runs instead)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long* tp = (long long*)lds;
    long long* tl = tp + DLRM_MAX_FEATURES;
    long long* tq = tl + DLRM_MAX_FEATURES;                 // gather mode only (the LDS is reserved either way)
    long long* to = tq + DLRM_MAX_FEATURES;
    long long* tr = to + DLRM_MAX_FEATURES;
    // images hold 2 NI rows (not 32): five waves' double buffers fit the 160 KiB for F <= 28.  The second 16-row tile then reads up to
    // four rows past its image — the next image or the selector slots behind the last one (always allocated): those rows only
    // feed outputs that are never stored (i, j >= F)
    constexpr int IMGB = 2 * NI * IDMA_D * 4;
    const int W = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
    char* img0 = (char*)(tr + DLRM_MAX_FEATURES) + (size_t)wave * 2 * IMGB;
    const unsigned img0_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)img0;

    table_to_lds(fa, tp, tl, F);
    if constexpr (GATHER) gather_to_lds(ga, tq, to, tr, F);
    for (int e = lane; e < 2 * IMGB / 16; e += 64) ((float4*)img0)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const long long b_stride = (long long)gridDim.x * W;
    long long b = (long long)blockIdx.x * W + wave;
    if (b >= B) return;
    DmaPlan pl;
    GatherCtx<NI> gc;
    // gather mode: the wave's ring of three selector slots lives behind the images
    const char* sel0 = (const char*)(tr + DLRM_MAX_FEATURES) + (size_t)W * 2 * IMGB + (size_t)wave * GSEL_WAVE_BYTES;
    const unsigned sel0_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)sel0;
    int sl = 0;                                             // slot of the current sample's selectors
    GatherSel gsn = {0u, 0u, 0u, 0u};                       // selectors of the NEXT sample, in registers
    if constexpr (GATHER) {
        gather_ctx_init<NI>(gc, tp, tl, tq, to, tr, F, lane, ga.idx_bits);
        const long long last = B - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long s = b + k * b_stride;
            gather_sel_issue<NI>(gc, s < B ? s : last, sel0_lds + k * GSEL_SLOT);
        }
        wait_vmcnt_i<0>();
        gather_rows_issue<NI>(gc, gather_sel_read(sel0, lane), b, lane, ga.idx_bits, ga.err, img0_lds);
        gsn = gather_sel_read(sel0 + GSEL_SLOT, lane);          // the next sample's selectors (all three slots have landed)
    } else {
        dma_plan_init(pl, tp, tl, F, lane, b, b_stride);
    }

    const int g = lane >> 4, li = lane & 15;
    const int P = (self & 1) ? F * (F + 1) / 2 : F * (F - 1) / 2;
    constexpr int NB = (2 * NI + 15) / 16;                  // 16-row tiles of the image (F <= 2 NI)
    const int dbg = self >> 6;                              // timing-only switches (interact_debug)
    // where this lane's four results of tile pair (r, c) go inside the R row (float index; -1 = not part of the output):
    // output row i = 16 r + 4 g + q, column j = 16 c + li — a function of the lane only, computed once
    int opos[NB * (NB + 1) / 2][4];
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * r + 4 * g + q, j = 16 * c + li;
                const bool ok = i < F && ((self & 1) ? (j <= i) : (j < i));
                opos[r * (r + 1) / 2 + c][q] = ok ? IDMA_D + pair_pos(i, j, F, self) : -1;
            }
    constexpr int NPAIR = NB * (NB + 1) / 2;
    int cur = 0;
    if constexpr (!GATHER) dma_issue<NI>(pl, img0_lds);
    for (; b < B; b += b_stride) {
        const bool more = b + b_stride < B && !(dbg & 1);
        if constexpr (GATHER) {
            // the next sample's selectors landed with the previous wait (they are older than the rows it retired): its rows go out
            // now, the selectors three samples ahead take the slot this sample's selectors just left, and the counted wait leaves
            // exactly those NI + 2 operations in flight while this sample is multiplied
            if (more) {
                const int sl1 = sl == 2 ? 0 : sl + 1, sl2 = sl1 == 2 ? 0 : sl1 + 1;
                gather_rows_issue<NI>(gc, gsn, b + b_stride, lane, ga.idx_bits, ga.err, img0_lds + (cur ^ 1) * IMGB);
                const long long s3 = b + 3 * b_stride;
                gather_sel_issue<NI>(gc, s3 < B ? s3 : B - 1, sel0_lds + sl * GSEL_SLOT);
                wait_vmcnt_i<NI + 2>();
                // the selectors two samples ahead are older than everything that wait leaves in flight: into registers now, their LDS round
                // trip rides with this sample's fragment reads
                gsn = gather_sel_read(sel0 + sl2 * GSEL_SLOT, lane);
                sl = sl1;
            } else if (!(dbg & 1)) wait_vmcnt_i<0>();
        } else {
            if (more) { dma_issue<NI>(pl, img0_lds + (cur ^ 1) * IMGB); wait_vmcnt_i<NI>(); }
            else if (!(dbg & 1)) wait_vmcnt_i<0>();
        }
        if (dbg & 2) { cur ^= 1; continue; }
        const char* my = img0 + cur * IMGB;
        // fragments of image rows li and 16 + li for the eight 16-wide k-steps: the three tile pairs of the lower triangle
        // ((0,0), (1,0), (1,1)) multiply these two row sets with each other, so 16 ds_read_b128 feed all 96 MFMAs of the sample
        float4 fr[NB][IDMA_D / 16];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int s = 0; s < IDMA_D / 16; ++s)
                fr[r][s] = *(const float4*)(my + (16 * r + li) * (IDMA_D * 4) + (((4 * s + g) ^ li) * 16));      // (row & 15) == li
        // (the scheduler sinks each k-step's two reads to just before its 12 MFMAs — one wave per SIMD, so those LDS round trips are exposed.
        // Pinning all reads in front of the first MFMA with a scheduling fence: round 5 measured the kernel 195 -> 188 us and the step no
        // faster (profiles/round5/interact_frags_first_ab.txt) and left it out; end of round 6: 0.196-0.198 -> 0.194 ms again, adopted below)
        // the tile pairs advance together, one k-substep at a time: consecutive MFMAs are independent, the two accumulators of a
        // pair (even / odd substeps, summed at the end — the summation order of every version of this kernel) are three issues apart
        // (x = image row 0, for R[:, 0:D]: read with the fragments, not as an exposed LDS round trip behind the MFMAs)
        const float4 xrow = *(const float4*)(my + (lane & 31) * 16);
#ifndef DLRM_FWD_PIN
#define DLRM_FWD_PIN 1
#endif
        if (DLRM_FWD_PIN) __builtin_amdgcn_sched_barrier(0);      // all fragment reads in front of the first MFMA (end of round 6: adopted)
        floatx4 acc[NPAIR][2];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) { acc[p][0] = (floatx4){0.f, 0.f, 0.f, 0.f}; acc[p][1] = (floatx4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < IDMA_D / 16; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int c = 0; c <= r; ++c) {
                        const float av = e == 0 ? fr[r][s].x : e == 1 ? fr[r][s].y : e == 2 ? fr[r][s].z : fr[r][s].w;
                        const float bv = e == 0 ? fr[c][s].x : e == 1 ? fr[c][s].y : e == 2 ? fr[c][s].z : fr[c][s].w;
                        acc[r * (r + 1) / 2 + c][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r * (r + 1) / 2 + c][e & 1], 0, 0, 0);
                    }
            }
        }
        float* Rb = R + b * ldr;
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const floatx4 sum = acc[p][0] + acc[p][1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (opos[p][q] >= 0 && !(dbg & 4)) Rb[opos[p][q]] = sum[q];
        }
        // R[:, 0:D] = x (row 0 of the image, un-swizzled: row & 15 == 0), then the alignment padding
        if (lane < 32) *(float4*)(Rb + 4 * lane) = xrow;
        for (long long d = IDMA_D + P + lane; d < ldr; d += 64) Rb[d] = 0.f;
        cur ^= 1;
    }
}


// backward, D = 128: dT = (dZ + dZ^T) · T per sample, T by LDS-DMA (same images as the forward kernel), the
// dR row by LDS-DMA too.  The symmetric S = dZ + dZ^T is never materialised: the 16x16x4 MFMA A fragment of a
// lane is S[16r + li][4kk + g], whose source position inside the dR row depends only on the lane -> 4*NB*NB LDS
// offsets computed once, 4*NB*NB ds_read_b32 per sample.  B fragments are ds_read_b128 of T rows: the four floats
// of a lane feed four MFMAs whose output column n = li then stands for d = 64*dq + 4*li + s, so a lane ends up
// with float4s of dT rows (16-byte, 256-B-per-row coalesced stores through the {pointer, stride} table).
constexpr int IDMA_DR_BYTES = 3072;      // dR row image (<= 656 floats for F = 32 with self pairs)

// UPD (dlrm_interact_bwd_gather_sgd, ABI 17): the sparse SGD step of the rows ONE lookup of the batch touches happens here — bit f - 1 of
// single_mask[b] says that gathered feature f of sample b is such a lookup (dlrm_emb_presort): its table row is staged in this wave's image
// anyway, so W[idx] = fma(-lr, dT, W[idx]) goes to the TABLE and the gradient row is not written (dlrm_emb_bwd_sgd_presorted skips the same
// lookups): per single lookup one row write instead of a row write + two row reads + a row write.  No other wave reads or writes that row
// (that is what single means), and this wave's read of it (the DMA of this sample's image) was retired before its MFMAs started.
template <int NI, bool GATHER, bool UPD = false>
__global__ __launch_bounds__(256) void interact_bwd_dma_kernel(FeatArgs fa, FeatArgs da, GatherArgs ga, long long B, int F, int self,
                                                               const float* __restrict__ dR, long long ldr,
                                                               const unsigned* __restrict__ single_mask = nullptr, DlrmStep neg_lr_ = DlrmStep{0.f, nullptr, 0.f}) {
    if (ga.pred.skip()) return;
    static_assert(!UPD || GATHER, "the fused update exists in gather mode only");
    constexpr int NB = (2 * NI + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long* tp = (long long*)lds;
    long long* tl = tp + DLRM_MAX_FEATURES;
    long long* dp = tl + DLRM_MAX_FEATURES;
    long long* dl = dp + DLRM_MAX_FEATURES;
    long long* tq = dl + DLRM_MAX_FEATURES;                 // gather mode only
    long long* to = tq + DLRM_MAX_FEATURES;
    long long* tr = to + DLRM_MAX_FEATURES;
    constexpr int DRB = GATHER ? GDR_BYTES : IDMA_DR_BYTES;      // dR row image
    char* img0 = (char*)(tr + DLRM_MAX_FEATURES) + (size_t)wave * (2 * IDMA_IMG + 2 * DRB);
    char* drow0 = img0 + 2 * IDMA_IMG;
    const unsigned img0_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)img0;
    const unsigned drow0_lds = img0_lds + 2 * IDMA_IMG;
    // gather mode: the wave's ring of three selector slots lives behind all the images
    const char* sel0 = (const char*)(tr + DLRM_MAX_FEATURES) + (size_t)4 * (2 * IDMA_IMG + 2 * DRB) + (size_t)wave * GSEL_WAVE_BYTES;
    const unsigned sel0_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)sel0;

    table_to_lds(fa, tp, tl, F);
    table_to_lds(da, dp, dl, F);
    if constexpr (GATHER) gather_to_lds(ga, tq, to, tr, F);
    for (int e = lane; e < (2 * IDMA_IMG + 2 * DRB) / 16; e += 64) ((float4*)img0)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const long long b_stride = (long long)gridDim.x * 4;
    long long b = (long long)blockIdx.x * 4 + wave;
    if (b >= B) return;
    DmaPlan pl;
    GatherCtx<NI> gc;
    if constexpr (GATHER) gather_ctx_init<NI>(gc, tp, tl, tq, to, tr, F, lane, ga.idx_bits, UPD ? single_mask : nullptr);
    else dma_plan_init(pl, tp, tl, F, lane, b, b_stride);
    // dR row: lane covers bytes [1024*c + 16*lane, +16) of the row, c < nr; lanes past the row pitch are masked
    const int nr = __builtin_amdgcn_readfirstlane((int)((ldr * 4 + 1023) / 1024));
    const char* dr_src = (const char*)(dR + b * ldr) + 16 * lane;
    const long long dr_step = b_stride * ldr * 4;
    auto issue_dr = [&](unsigned dst) {
        for (int c = 0; c < nr; ++c)
            if ((long long)(1024 * c + 16 * lane) < ldr * 4) glds16_v(dr_src + 1024 * c, dst + c * 1024);
        dr_src += dr_step;
    };

    const int g = lane >> 4, li = lane & 15;
    // A-fragment sources inside the dR row (float index; -1 = structural zero), doubled on the diagonal when self pairs exist
    int a_off[NB][4 * NB];
    float a_scale[NB][4 * NB];
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int kk = 0; kk < 4 * NB; ++kk) {
            const int i = 16 * r + li, j = 4 * kk + g;
            // structural zeros read the LAST word of the dR image: the row's DMA never reaches it (the launcher requires ldr * 4 < DRB) and it was
            // zeroed with the images — so a fragment is one multiply (x 1, x 2 on the diagonal), not a compare + multiply + select, and a
            // structural zero stays zero whatever dR holds
            int off = DRB / 4 - 1; float sc = 1.f;
            if (i < F && j < F) {
                if (i == j) { if (self & 1) { off = IDMA_D + pair_pos(i, i, F, self); sc = 2.f; } }
                else {
                    const int hi = i > j ? i : j, lo = i > j ? j : i;
                    off = IDMA_D + pair_pos(hi, lo, F, self);
                }
            }
            a_off[r][kk] = off * 4; a_scale[r][kk] = sc;
        }
    // destination rows of this lane: i = 16r + 4g + q.  GLOBAL pointers: as generic ones (rounds 2-6) their stores were flat_store_dwordx4, which
    // tick lgkmcnt as well — the next sample's first LDS read then waited for the previous sample's row stores to be acknowledged.
    // (UPD computes its destinations per sample — table row or gradient row — and keeps only which rows exist: 32 registers fewer)
    gchar* orow[NB][4];
    long long ostep[NB][4];
    unsigned rowbits = 0u;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * r + 4 * g + q;
            const int ii = i < F ? i : 0;
            if (i < F) rowbits |= 1u << (4 * r + q);
            if constexpr (!UPD) {
                orow[r][q] = (i < F) ? (gchar*)dp[ii] + (b * dl[ii] + 4 * li) * 4 : nullptr;
                ostep[r][q] = b_stride * dl[ii] * 4;
            }
        }
    // how many row-store instructions a sample issues (one exists iff some lane group has the row: g = 0); (UPD) which of this lane's eight
    // rows are gathered features at all (bit 4 r + q) and the step size
    unsigned tabbits = 0u;
    float neg_lr = 0.f;
    int n_st = 0;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * r + 4 * g + q;
            if (UPD && i >= 1 && i < F && tr[i] >= 0) tabbits |= 1u << (4 * r + q);       // (plain features have rows = -1)
            n_st += (16 * r + q < F) ? IDMA_D / 64 : 0;
        }
    n_st = __builtin_amdgcn_readfirstlane(n_st);
    if constexpr (UPD) neg_lr = neg_lr_;
#ifndef DLRM_BWD_STORES_IN_FLIGHT
#define DLRM_BWD_STORES_IN_FLIGHT 1
#endif
#ifndef DLRM_UPD_DIAG
#define DLRM_UPD_DIAG 0
#endif

    int cur = 0, sl = 0;
    bool stores_behind = false;        // (UPD) the queue holds a sample's row stores behind the current sample's loads
    GatherSel gsn = {0u, 0u, 0u, 0u};  // selectors of the NEXT sample, in registers (see the forward kernel)
    if constexpr (GATHER) {
        const long long last = B - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long s = b + k * b_stride;
            gather_sel_issue<NI>(gc, s < B ? s : last, sel0_lds + k * GSEL_SLOT);
        }
        wait_vmcnt_i<0>();
        gather_rows_issue<NI>(gc, gather_sel_read(sel0, lane), b, lane, ga.idx_bits, ga.err, img0_lds);
        gsn = gather_sel_read(sel0 + GSEL_SLOT, lane);
    } else {
        dma_issue<NI>(pl, img0_lds);
    }
    issue_dr(drow0_lds);
    for (; b < B; b += b_stride) {
        const bool more = b + b_stride < B;
        // (UPD) which of this lane's rows are single lookups, and where their table rows live: read from this sample's selector slot before
        // the slot is handed to the selectors three samples ahead — but BEHIND the next sample's row loads (their issue must not wait for these
        // LDS round trips and the address arithmetic).  dst = the table row (gathered rows are D floats apart) or the gradient row; one store
        // per row either way
        gchar* dst[NB][4];
        unsigned ones = 0u;
        auto upd_select = [&]() {
            const char* slot = sel0 + sl * GSEL_SLOT;
            unsigned um = *(const unsigned*)(slot + GSEL_WORD_OFF);
            if (DLRM_UPD_DIAG == 1) um = 0u;
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * r + 4 * g + q;
                    const unsigned id = *(const unsigned*)(slot + 4 * (i & 31));
                    const bool one = ((tabbits >> (4 * r + q)) & 1u) && ((um >> ((i - 1) & 31)) & 1u);
                    gchar* wr = (gchar*)tp[i & 31] + 16 * li + ((unsigned long long)id << 9);
                    // (32 x 32 -> 64-bit multiply-add: one v_mad_u64_u32; the entry point bounds B and the row pitches)
                    gchar* gr = (gchar*)dp[i & 31] + 16 * li + (unsigned long long)(unsigned)b * ((unsigned)dl[i & 31] * 4u);
                    dst[r][q] = (one && DLRM_UPD_DIAG != 2) ? wr : gr;
                    ones |= one ? 1u << (4 * r + q) : 0u;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if constexpr (GATHER) {
            // same schedule as the forward kernel: rows + dR row of the next sample and the selectors three samples ahead stay in
            // flight (NI + nr + 2 operations) while this sample is multiplied
            if (more) {
                const int sl1 = sl == 2 ? 0 : sl + 1;
                gather_rows_issue<NI>(gc, gsn, b + b_stride, lane, ga.idx_bits, ga.err, img0_lds + (cur ^ 1) * IDMA_IMG);
                issue_dr(drow0_lds + (cur ^ 1) * DRB);
                if constexpr (UPD) upd_select();
                const long long s3 = b + 3 * b_stride;
                gather_sel_issue<NI>(gc, s3 < B ? s3 : B - 1, sel0_lds + sl * GSEL_SLOT);
                // (the previous sample's row stores — issued after this sample's loads, before the next one's — may stay in flight too: vmcnt
                // retires in issue order)
                if (DLRM_BWD_STORES_IN_FLIGHT && stores_behind) wait_vmcnt_rt(NI + nr + 2 + n_st);
                else wait_vmcnt_rt(NI + nr + 2);
                gsn = gather_sel_read(sel0 + (sl1 == 2 ? 0 : sl1 + 1) * GSEL_SLOT, lane);      // two samples ahead: landed (older than what the wait leaves)
                sl = sl1;
            } else {
                if constexpr (UPD) upd_select();
                wait_vmcnt_i<0>();
            }
        } else if (more) {
            dma_issue<NI>(pl, img0_lds + (cur ^ 1) * IDMA_IMG);
            issue_dr(drow0_lds + (cur ^ 1) * DRB);
            if (DLRM_BWD_STORES_IN_FLIGHT && stores_behind) wait_vmcnt_rt(NI + nr + n_st);
            else wait_vmcnt_rt(NI + nr);
        } else wait_vmcnt_i<0>();
        const char* my = img0 + cur * IDMA_IMG;
        const char* dr = drow0 + cur * DRB;
        float aS[NB][4 * NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int kk = 0; kk < 4 * NB; ++kk) {
                const float v = *(const float*)(dr + a_off[r][kk]);
                aS[r][kk] = a_scale[r][kk] * v;
            }
        // (B fragments of both 64-column halves read up front: measured, no effect — 0.315-0.322 vs 0.317-0.323 ms)
#pragma unroll
        for (int dq = 0; dq < IDMA_D / 64; ++dq) {
            float4 bT[4 * NB];
#pragma unroll
            for (int kk = 0; kk < 4 * NB; ++kk) {
                const int jr = 4 * kk + g;                 // rows >= F of the image are zero
                bT[kk] = *(const float4*)(my + jr * (IDMA_D * 4) + (((16 * dq + li) ^ (jr & 15)) * 16));
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                // (UPD) the table rows this lane may have to step, read HERE — in front of the block's 32 MFMAs, not one LDS round trip per
                // store behind them (one wave per SIMD: nothing else hides a read's latency; sixteen exposed reads per sample cost 1 us
                // per sample = 63 us per launch)
                float4 wv[4];
                if constexpr (UPD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // image row i, this lane's quad, un-swizzled — or, for a row this lane does not step, the image's zero row 31 (F <= 27:
                        // rows F..31 are zeroed once and never written): the store below is then ONE fma per element for every row,
                        // fma(-lr, dT, W) or fma(1, dT, 0) = dT, instead of an fma and a select per element
                        const int i = ((ones >> (4 * r + q)) & 1u) ? 16 * r + 4 * g + q : 31;
                        wv[q] = *(const float4*)(my + i * (IDMA_D * 4) + (((16 * dq + li) ^ (i & 15)) * 16));
                    }
                }
                // feature 0's two extra operands (the x part of dR; x itself for the ReLU derivative), read HERE by every lane: inside the
                // lane-divergent branch behind the MFMAs each was an exposed LDS round trip (one wave per SIMD)
                float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), y0 = x0;
                if (r == 0) {
                    x0 = *(const float4*)(dr + (64 * dq + 4 * li) * 4);
                    y0 = *(const float4*)(my + (16 * dq + li) * 16);
                }
                floatx4 acc[4];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) acc[s_] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4 * NB; ++kk) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk], bT[kk].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk], bT[kk].y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk], bT[kk].z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[r][kk], bT[kk].w, acc[3], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((rowbits >> (4 * r + q)) & 1u) {
                        float4 v = make_float4(acc[0][q], acc[1][q], acc[2][q], acc[3][q]);
                        if (r == 0 && q == 0 && g == 0) {       // feature 0 also feeds R[:, 0:D]
                            const float4 x = x0;
                            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                            if (self & 4) {                     // feature 0 is a ReLU output: its derivative is applied here (image row 0 = x)
                                const float4 y = y0;
                                v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f;
                                v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
                            }
                        }
                        if constexpr (UPD) {
                            // a single lookup: the SGD step itself goes to the table row
                            const float4 w = wv[q];
                            const float sc = ((ones >> (4 * r + q)) & 1u) ? neg_lr : 1.f;
                            v.x = __builtin_fmaf(sc, v.x, w.x); v.y = __builtin_fmaf(sc, v.y, w.y);
                            v.z = __builtin_fmaf(sc, v.z, w.z); v.w = __builtin_fmaf(sc, v.w, w.w);
                            *(gfloatx4*)(dst[r][q] + dq * 256) = (floatx4){v.x, v.y, v.z, v.w};
                        } else
                            *(gfloatx4*)(orow[r][q] + dq * 256) = (floatx4){v.x, v.y, v.z, v.w};
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!UPD && ((rowbits >> (4 * r + q)) & 1u)) orow[r][q] += ostep[r][q];
        cur ^= 1;
        stores_behind = true;
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

static bool interact_dma_ok(int F, int D, int vec) {
    static const int off = DLRM_TUNE_ENV("DLRM_INTERACT_PATH", 0) == 1;      // tuning builds: 1 forces the register-staged kernels
    return !off && D == IDMA_D && F >= 1 && F <= IDMA_ROWS && vec;
}

// tuning aid (env DLRM_INTERACT_DEBUG, D = 128 forward kernels): 1 = no DMA and no waits in the sample loop (compute + stores only),
// 2 = no multiplication and no stores (DMA only), 4 = no stores — WRONG results, timing only
static int interact_debug() {
    static int dbg = -1;
    if (dbg < 0) dbg = DLRM_DEBUG_ENV("DLRM_INTERACT_DEBUG", 7);
    return dbg;
}

// forward D = 128 kernels: waves per workgroup (one workgroup per CU) and its LDS — tables, two 2 NI-row images per wave, the selector
// slots (gather mode; in plain mode they are the slack the second 16-row tile may read into)
static size_t fwd_dma_lds(int ni, int W) {
    // a 16-row fragment tile of the LAST wave's last image reads up to (16 * tiles - 2 NI) rows of 512 B past the image: what lies behind
    // it are the W selector slots and this slack — sized from the over-read, not assumed (with DLRM_INTERACT_WAVES <= 3 and few features
    // the selector slots alone were smaller than the over-read: out-of-bounds LDS reads whose values were discarded)
    const size_t over = (size_t)(16 * ((2 * ni + 15) / 16) - 2 * ni) * IDMA_D * 4, sel = (size_t)W * GSEL_WAVE_BYTES;
    const size_t slack = (over > sel && over - sel > 2048) ? over - sel : 2048;
    return 5 * DLRM_MAX_FEATURES * sizeof(long long) + (size_t)W * (2 * (size_t)(2 * ni * IDMA_D * 4) + GSEL_WAVE_BYTES) + slack;
}
static int fwd_dma_waves(int ni) {
    static const int forced = DLRM_TUNE_ENV("DLRM_INTERACT_WAVES", 0);      // (tuning builds)
    int W = (forced >= 1 && forced <= 5) ? forced : 4;      // five waves fit F <= 28 but two of them then share a SIMD: 245 vs 211 us measured
    while (W > 1 && fwd_dma_lds(ni, W) > 160 * 1024) --W;
    return W;
}

static int pick_grid(int64_t B) {
    // 4 samples per workgroup pass; ~2 resident workgroups per CU x 256 CUs x 4 oversubscription
    int64_t nb = (B + 3) / 4;
    if (nb > 2048) nb = 2048;
    return (int)nb;
}

}  // namespace

static int fill_gather(GatherArgs& ga, int F, const void* const* gidx, const void* const* goff, const int64_t* grows, int idx_bits,
                       int64_t* err) {
    ga.err = (long long*)err; ga.idx_bits = idx_bits;
    ga.pred.flag = nullptr; ga.pred.nonzero = 0;
    const void* any_idx = nullptr; const void* any_off = nullptr;
    for (int f = 0; gidx && f < F; ++f) if (gidx[f]) { any_idx = gidx[f]; any_off = goff ? goff[f] : nullptr; break; }
    for (int f = 0; f < DLRM_MAX_FEATURES; ++f) {
        const bool g = gidx && f < F && gidx[f];
        if (g && (!goff || !goff[f] || !grows || grows[f] <= 0)) return DLRM_E_ARG;
        if (g && grows[f] > 0xFFFFFFFFLL) return DLRM_E_RANGE;          // row selectors travel as 32-bit values inside the kernels
        // plain features: a valid dummy pointer (their loads are unconditional and ignored) and rows = -1
        ga.idx[f] = g ? gidx[f] : any_idx; ga.off[f] = g ? goff[f] : any_off; ga.rows[f] = g ? grows[f] : -1;
    }
    return 0;
}

// F <= 27: the dR row (D + F (F + 1) / 2 floats with self pairs, padded to 4) must fit the 2 KiB image of the gather backward
extern "C" int dlrm_interact_gather_ok(int F, int D) {
    return (D == IDMA_D && F >= 1 && F <= IDMA_ROWS && 4 * ((IDMA_D + F * (F + 1) / 2 + 3) & ~3) <= GDR_BYTES) ? 1 : 0;
}

static int interact_fwd_impl(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* gidx, const void* const* goff, const int64_t* grows, int idx_bits,
                             int self_interaction, float* R, int64_t ldr, int64_t* err, void* stream, DlrmPred pred = DlrmPred{nullptr, 0});

extern "C" int dlrm_interact_fwd(int64_t B, int F, int D, const void* const* feat_host,
                                 const int64_t* feat_ld_host, int self_interaction, float* R,
                                 int64_t ldr, void* stream) {
    return interact_fwd_impl(B, F, D, feat_host, feat_ld_host, nullptr, nullptr, nullptr, 64, self_interaction, R, ldr, nullptr, stream);
}

extern "C" int dlrm_interact_fwd_gather(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                        const void* const* index_host, const void* const* offsets_host,
                                        const int64_t* rows_host, int idx_bits, int self_interaction, float* R, int64_t ldr,
                                        int64_t* err, void* stream) {
    if (!index_host || !offsets_host || !rows_host) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    return interact_fwd_impl(B, F, D, feat_host, feat_ld_host, index_host, offsets_host, rows_host, idx_bits, self_interaction, R,
                             ldr, err, stream);
}

// One call, two implementations, chosen ON THE DEVICE (ABI 16): the same arguments as dlrm_interact_fwd / _fwd_gather / _bwd / _bwd_gather plus a
// predicate — the launch's workgroups return at once unless (*pred_flag != 0) == (pred_nonzero != 0).  Only the D = 128 LDS-DMA kernels take
// one (the shapes of the fused path); other shapes: DLRM_E_MODE.
extern "C" int dlrm_interact_fwd_pred(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                      const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                                      int idx_bits, int self_interaction, float* R, int64_t ldr, int64_t* err,
                                      const int32_t* pred_flag, int pred_nonzero, void* stream) {
    if (index_host && (!offsets_host || !rows_host)) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    return interact_fwd_impl(B, F, D, feat_host, feat_ld_host, index_host, offsets_host, rows_host, idx_bits, self_interaction, R,
                             ldr, err, stream, DlrmPred{(const int*)pred_flag, pred_nonzero});
}

static int interact_fwd_impl(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* gidx, const void* const* goff, const int64_t* grows, int idx_bits,
                             int self_interaction, float* R, int64_t ldr, int64_t* err, void* stream, DlrmPred pred) {
    if (B <= 0 || F <= 0 || D <= 0 || !feat_host || !feat_ld_host || !R) return DLRM_E_ARG;
    if (F > DLRM_MAX_FEATURES) {
        fprintf(stderr, "libdlrm_hip: dlrm_interact_fwd: F=%d exceeds %d features\n", F, DLRM_MAX_FEATURES);
        return DLRM_E_RANGE;
    }
    if (self_interaction < 0 || self_interaction > 2) return DLRM_E_MODE;     // 0 tril, 1 tril + diagonal, 2 torchrec triu order
    const int P = (self_interaction & 1) ? F * (F + 1) / 2 : F * (F - 1) / 2;
    if (ldr < D + P) return DLRM_E_ARG;
    FeatArgs fa;
    int rc = fill_feat(fa, F, feat_host, feat_ld_host);
    if (rc) return rc;
    int vec = (D % 4 == 0);
    for (int f = 0; f < F; ++f) vec = vec && dlrm_aligned16(feat_host[f]) && (feat_ld_host[f] % 4 == 0);
    GatherArgs ga;
    rc = fill_gather(ga, F, gidx, goff, grows, idx_bits, err);
    if (rc) return rc;
    ga.pred = pred;
    if (gidx) {          // gathered features exist only in the D = 128 LDS-DMA kernel
        if (!(dlrm_interact_gather_ok(F, D) && vec && dlrm_aligned16(R) && ldr % 4 == 0)) return DLRM_E_MODE;
        const int ni = (F + 1) / 2;
        const int W = fwd_dma_waves(ni);
        const size_t lds = fwd_dma_lds(ni, W);
        int64_t nb = (B + W - 1) / W; if (nb > 256) nb = 256;
#define FWD_G(NIV)                                                                                           \
        do {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)interact_fwd_dma_kernel<NIV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((interact_fwd_dma_kernel<NIV, true>), dim3((unsigned)nb), dim3(64 * W), lds, (hipStream_t)stream, fa, ga, \
                               (long long)B, F, (self_interaction & 3) | (interact_debug() << 6), R, (long long)ldr); \
        } while (0)
        switch (ni) {
            case 1: FWD_G(1); break;   case 2: FWD_G(2); break;   case 3: FWD_G(3); break;   case 4: FWD_G(4); break;
            case 5: FWD_G(5); break;   case 6: FWD_G(6); break;   case 7: FWD_G(7); break;   case 8: FWD_G(8); break;
            case 9: FWD_G(9); break;   case 10: FWD_G(10); break; case 11: FWD_G(11); break; case 12: FWD_G(12); break;
            case 13: FWD_G(13); break; case 14: FWD_G(14); break; case 15: FWD_G(15); break; default: FWD_G(16); break;
        }
#undef FWD_G
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    if (interact_dma_ok(F, D, vec) && dlrm_aligned16(R) && ldr % 4 == 0) {
        const int ni = (F + 1) / 2;
        const int W = fwd_dma_waves(ni);
        const size_t lds = fwd_dma_lds(ni, W);                                                      // one workgroup per CU
        int64_t nb = (B + W - 1) / W; if (nb > 256) nb = 256;
#define FWD_DMA(NIV)                                                                                         \
        do {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)interact_fwd_dma_kernel<NIV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((interact_fwd_dma_kernel<NIV, false>), dim3((unsigned)nb), dim3(64 * W), lds, (hipStream_t)stream, fa, ga, \
                               (long long)B, F, (self_interaction & 3) | (interact_debug() << 6), R, (long long)ldr); \
        } while (0)
        switch (ni) {
            case 1: FWD_DMA(1); break;   case 2: FWD_DMA(2); break;   case 3: FWD_DMA(3); break;   case 4: FWD_DMA(4); break;
            case 5: FWD_DMA(5); break;   case 6: FWD_DMA(6); break;   case 7: FWD_DMA(7); break;   case 8: FWD_DMA(8); break;
            case 9: FWD_DMA(9); break;   case 10: FWD_DMA(10); break; case 11: FWD_DMA(11); break; case 12: FWD_DMA(12); break;
            case 13: FWD_DMA(13); break; case 14: FWD_DMA(14); break; case 15: FWD_DMA(15); break; default: FWD_DMA(16); break;
        }
#undef FWD_DMA
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    if (pred.flag) return DLRM_E_MODE;                         // (predicated launches exist for the LDS-DMA kernels only)
    const int Dp = (D + 15) & ~15;
    const size_t lds = 2 * DLRM_MAX_FEATURES * sizeof(long long) + 4 * (size_t)(((F + 15) >> 4) * 16) * (Dp + 4) * sizeof(float);
    if (lds > 160 * 1024) {
        fprintf(stderr, "libdlrm_hip: dlrm_interact_fwd: F=%d, D=%d needs %zu B of LDS (> 160 KiB)\n", F, D, lds);
        return DLRM_E_RANGE;
    }
    (void)hipFuncSetAttribute((const void*)interact_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(interact_fwd_kernel, dim3(pick_grid(B)), dim3(256), lds, (hipStream_t)stream, fa,
                       (long long)B, F, D, self_interaction & 3, R, (long long)ldr, vec,
                       vec ? log2_exact(D / 4) : -1);
    DLRM_LAUNCH_CHECK();
    return 0;
}

static int interact_bwd_impl(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* gidx, const void* const* goff, const int64_t* grows, int idx_bits,
                             int self_interaction, const float* dR, int64_t ldr, void* const* dfeat_host,
                             const int64_t* dfeat_ld_host, int64_t* err, void* stream, DlrmPred pred = DlrmPred{nullptr, 0},
                             const uint32_t* single_mask = nullptr, DlrmStep neg_lr = DlrmStep{0.f, nullptr, 0.f});

extern "C" int dlrm_interact_bwd(int64_t B, int F, int D, const void* const* feat_host,
                                 const int64_t* feat_ld_host, int self_interaction, const float* dR,
                                 int64_t ldr, void* const* dfeat_host, const int64_t* dfeat_ld_host,
                                 void* stream) {
    return interact_bwd_impl(B, F, D, feat_host, feat_ld_host, nullptr, nullptr, nullptr, 64, self_interaction, dR, ldr, dfeat_host,
                             dfeat_ld_host, nullptr, stream);
}

extern "C" int dlrm_interact_bwd_gather(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                        const void* const* index_host, const void* const* offsets_host,
                                        const int64_t* rows_host, int idx_bits, int self_interaction, const float* dR,
                                        int64_t ldr, void* const* dfeat_host, const int64_t* dfeat_ld_host, int64_t* err,
                                        void* stream) {
    if (!index_host || !offsets_host || !rows_host) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    return interact_bwd_impl(B, F, D, feat_host, feat_ld_host, index_host, offsets_host, rows_host, idx_bits, self_interaction, dR,
                             ldr, dfeat_host, dfeat_ld_host, err, stream);
}

extern "C" int dlrm_interact_bwd_pred(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                      const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                                      int idx_bits, int self_interaction, const float* dR, int64_t ldr, void* const* dfeat_host,
                                      const int64_t* dfeat_ld_host, int64_t* err, const int32_t* pred_flag, int pred_nonzero, void* stream) {
    if (index_host && (!offsets_host || !rows_host)) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    return interact_bwd_impl(B, F, D, feat_host, feat_ld_host, index_host, offsets_host, rows_host, idx_bits, self_interaction, dR,
                             ldr, dfeat_host, dfeat_ld_host, err, stream, DlrmPred{(const int*)pred_flag, pred_nonzero});
}

// ABI 17: dlrm_interact_bwd_pred (gather form) that ALSO takes the sparse SGD step of the single lookups (see interact_bwd_dma_kernel<.., UPD>)
extern "C" int dlrm_interact_bwd_gather_sgd(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                                            const void* const* index_host, const void* const* offsets_host, const int64_t* rows_host,
                                            int idx_bits, int self_interaction, const float* dR, int64_t ldr, void* const* dfeat_host,
                                            const int64_t* dfeat_ld_host, const uint32_t* single_mask, float lr, const float* lr_dev, int64_t* err,
                                            const int32_t* pred_flag, int pred_nonzero, void* stream) {
    if (!index_host || !offsets_host || !rows_host || !single_mask) return DLRM_E_ARG;
    if (idx_bits != 32 && idx_bits != 64) return DLRM_E_MODE;
    // bit f - 1 of a mask word names feature f: features 1 .. F-1 must all be tables (feature 0 the dense block), rows of D floats
    if (index_host[0]) return DLRM_E_ARG;
    for (int f = 1; f < F; ++f) if (!index_host[f] || feat_ld_host[f] != D) return DLRM_E_ARG;
    if (B >= ((int64_t)1 << 32)) return DLRM_E_RANGE;                    // (gradient-row addresses: sample x pitch as a 32 x 32-bit product)
    for (int f = 0; f < F; ++f) if (dfeat_ld_host && (dfeat_ld_host[f] < 0 || dfeat_ld_host[f] >= ((int64_t)1 << 30))) return DLRM_E_RANGE;
    return interact_bwd_impl(B, F, D, feat_host, feat_ld_host, index_host, offsets_host, rows_host, idx_bits, self_interaction, dR,
                             ldr, dfeat_host, dfeat_ld_host, err, stream, DlrmPred{(const int*)pred_flag, pred_nonzero}, single_mask,
                             dlrm_step_neg(lr, lr_dev));
}

static int interact_bwd_impl(int64_t B, int F, int D, const void* const* feat_host, const int64_t* feat_ld_host,
                             const void* const* gidx, const void* const* goff, const int64_t* grows, int idx_bits,
                             int self_interaction, const float* dR, int64_t ldr, void* const* dfeat_host,
                             const int64_t* dfeat_ld_host, int64_t* err, void* stream, DlrmPred pred, const uint32_t* single_mask,
                             DlrmStep neg_lr) {
    if (B <= 0 || F <= 0 || D <= 0 || !feat_host || !feat_ld_host || !dR || !dfeat_host || !dfeat_ld_host)
        return DLRM_E_ARG;
    if (F > DLRM_MAX_FEATURES) return DLRM_E_RANGE;
    // bits 0-1: 0 tril, 1 tril + diagonal, 2 torchrec triu order; bit 2 (DLRM_INTERACT_RELU_X, backward only): feature 0 is the output of a
    // ReLU and dfeat_0 is multiplied by its derivative [feature 0 > 0] (the bottom tower's last act_bwd pass, dlrm_s_pytorch.py:238-241)
    if (self_interaction < 0 || self_interaction > 7 || (self_interaction & 3) > 2) return DLRM_E_MODE;
    const int P = (self_interaction & 1) ? F * (F + 1) / 2 : F * (F - 1) / 2;
    if (ldr < D + P) return DLRM_E_ARG;
    FeatArgs fa, da;
    int rc = fill_feat(fa, F, feat_host, feat_ld_host);
    if (rc) return rc;
    rc = fill_feat(da, F, (const void* const*)dfeat_host, dfeat_ld_host);
    if (rc) return rc;
    int vec = (D % 4 == 0);
    for (int f = 0; f < F; ++f) vec = vec && dlrm_aligned16(feat_host[f]) && (feat_ld_host[f] % 4 == 0);
    {
        int dvec = 1;
        for (int f = 0; f < F; ++f) dvec = dvec && dlrm_aligned16(dfeat_host[f]) && (dfeat_ld_host[f] % 4 == 0);
        GatherArgs ga;
        rc = fill_gather(ga, F, gidx, goff, grows, idx_bits, err);
        if (rc) return rc;
        ga.pred = pred;
        const bool dma_path = (gidx ? (dlrm_interact_gather_ok(F, D) && vec) : interact_dma_ok(F, D, vec)) && dvec && dlrm_aligned16(dR) &&
                              ldr % 4 == 0 && ldr * 4 < (gidx ? GDR_BYTES : IDMA_DR_BYTES);      // (strictly: the image's last word stays zero)
        if (gidx && !dma_path) return DLRM_E_MODE;
        if (single_mask && !gidx) return DLRM_E_ARG;
        if (dma_path) {
            const size_t lds_dma = 7 * DLRM_MAX_FEATURES * sizeof(long long) +
                                   (gidx ? 4 * (2 * (size_t)IDMA_IMG + 2 * (size_t)GDR_BYTES + (size_t)GSEL_WAVE_BYTES)
                                         : 4 * (2 * (size_t)IDMA_IMG + 2 * (size_t)IDMA_DR_BYTES));
            int64_t nb = (B + 3) / 4; if (nb > 256) nb = 256;
            const int ni = (F + 1) / 2;
#define BWD_DMA(NIV)                                                                                         \
            do {                                                                                             \
                if (gidx && single_mask) {                                                                   \
                    (void)hipFuncSetAttribute((const void*)interact_bwd_dma_kernel<NIV, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma); \
                    hipLaunchKernelGGL((interact_bwd_dma_kernel<NIV, true, true>), dim3((unsigned)nb), dim3(256), lds_dma, (hipStream_t)stream, fa, da, ga, \
                                       (long long)B, F, self_interaction & 7, dR, (long long)ldr, (const unsigned*)single_mask, neg_lr); \
                } else if (gidx) {                                                                           \
                    (void)hipFuncSetAttribute((const void*)interact_bwd_dma_kernel<NIV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma); \
                    hipLaunchKernelGGL((interact_bwd_dma_kernel<NIV, true>), dim3((unsigned)nb), dim3(256), lds_dma, (hipStream_t)stream, fa, da, ga, \
                                       (long long)B, F, self_interaction & 7, dR, (long long)ldr);       \
                } else {                                                                                     \
                    (void)hipFuncSetAttribute((const void*)interact_bwd_dma_kernel<NIV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma); \
                    hipLaunchKernelGGL((interact_bwd_dma_kernel<NIV, false>), dim3((unsigned)nb), dim3(256), lds_dma, (hipStream_t)stream, fa, da, ga, \
                                       (long long)B, F, self_interaction & 7, dR, (long long)ldr);       \
                }                                                                                            \
            } while (0)
            switch (ni) {
                case 1: BWD_DMA(1); break;   case 2: BWD_DMA(2); break;   case 3: BWD_DMA(3); break;   case 4: BWD_DMA(4); break;
                case 5: BWD_DMA(5); break;   case 6: BWD_DMA(6); break;   case 7: BWD_DMA(7); break;   case 8: BWD_DMA(8); break;
                case 9: BWD_DMA(9); break;   case 10: BWD_DMA(10); break; case 11: BWD_DMA(11); break; case 12: BWD_DMA(12); break;
                case 13: BWD_DMA(13); break; case 14: BWD_DMA(14); break; case 15: BWD_DMA(15); break; default: BWD_DMA(16); break;
            }
#undef BWD_DMA
            DLRM_LAUNCH_CHECK();
            return 0;
        }
    }
    if (pred.flag) return DLRM_E_MODE;                         // (predicated launches exist for the LDS-DMA kernels only)
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
                           self_interaction & 7, dR, (long long)ldr, vec, d4s);                      \
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
