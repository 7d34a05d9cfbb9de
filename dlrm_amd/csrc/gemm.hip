// gemm.hip — fp32 MFMA GEMM with fused MLP epilogues for gfx950 (the bottom/top MLP towers).
//
// Reference call sites replaced: nn.Linear + nn.ReLU / nn.Sigmoid inside the nn.Sequential built
// by DLRM_Net.create_mlp and run by apply_mlp (dlrm_s_pytorch.py:208-246, 399-405), plus their
// autograd (AddmmBackward / ReluBackward / SigmoidBackward) in E.backward() (:1613).
//
// One kernel template, C[M,N] = sum_k A(m,k)·B(n,k), instantiated for the three operand layouts the
// three MLP GEMMs need, so that every operand tile is a straight row copy global -> LDS:
//   forward  Y  = X·W^T   : A = X  (k contiguous)      B = W (k contiguous)
//   dgrad    dX = dY·W    : A = dY (k contiguous)      B = W (k = row index: "k strided")
//   wgrad    dW = dY^T·X  : A = dY (k = row, strided)  B = X (k = row, strided), split over k = batch
//
// Tile 128x128x32, 256 threads = 4 waves in a 2x2 grid, each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 accumulator registers).  The k order inside a tile is
// permuted so that the 4 k-values a lane feeds to 4 consecutive MFMAs are contiguous: a
// k-contiguous operand fragment is ONE ds_read_b128 (rows padded to 36 floats: conflict-free),
// a k-strided fragment is 4 conflict-free ds_read_b32.  Global -> register -> LDS staging is
// double buffered with one barrier per k-tile; the next tile's global loads are issued before the
// MFMAs of the current one.  Workgroup ids are remapped so that the n-tiles sharing an A row panel
// run on the same XCD (private L2).
//
// Fused: +bias and ReLU/sigmoid (forward epilogue); activation-derivative mask of the previous layer (dgrad
// epilogue); bias gradient as row sums of the dY^T fragments inside the wgrad main loop; atomic accumulation of
// the batch splits (wgrad).  Accumulators hold the TRANSPOSED sub-tiles so that the epilogue moves 16 bytes
// per lane through a wave-private LDS staging area and writes/reads whole 256-byte row segments.
#include <stdlib.h>
#include <string.h>
#include "common.h"
// fragment-read schedule (gemm3_kernel SCHED: 0 one register set, 1 both halves of a k-tile before its first product) of the forward /
// data-gradient / weight-gradient forms of the native fp32 GEMM
#ifndef DLRM_SCHED_FWD
#define DLRM_SCHED_FWD 0
#endif
#ifndef DLRM_SCHED_DGRAD
#define DLRM_SCHED_DGRAD 0
#endif
#ifndef DLRM_SCHED_WGRAD
#define DLRM_SCHED_WGRAD 1
#endif
#ifndef DLRM_WGRAD_MAXSPLITS_DEFAULT
#define DLRM_WGRAD_MAXSPLITS_DEFAULT 64
#endif
#ifndef DLRM_WGRAD_WGS_DEFAULT
#define DLRM_WGRAD_WGS_DEFAULT 1024
#endif
// stages of the LDS ring of the 128-row fp32 tiles (gemm3_kernel NST) per form
#ifndef DLRM_NST_FWD
#define DLRM_NST_FWD 2
#endif
#ifndef DLRM_NST_DGRAD
#define DLRM_NST_DGRAD 2
#endif
#ifndef DLRM_NST_WGRAD
#define DLRM_NST_WGRAD 2
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_KC = BK + 4;    // k-contiguous operand tile: [128][36] floats
constexpr int LD_KS = 128 + 4;   // k-strided operand tile  : [32][132] floats
constexpr int TILE_F = 128 * LD_KC;          // 4608 floats (== 32 * 144 > 32 * 132, shared size)
static_assert(TILE_F >= BK * LD_KS, "tile buffer too small");

struct GemmArgs {
    long long M, N, K;             // GEMM extents (M x N output, K reduction)
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    int vecA, vecB;                // 16-byte vector loads legal for the operand
    long long kchunk;              // reduction length per blockIdx.z slice (multiple of BK)
    const float* bias;             // [N] added before activation            (nullable)
    int act;                       // DLRM_ACT_* applied to the result
    const float* mask; long long ldmask; int mask_act;   // result *= act'(mask[m,n])   (nullable)
    float* rowsumA;                // [M] += row sums of the A operand over this k-slice (wgrad: bias gradient)  (nullable)
    long long rowsum_split_stride; // > 0: k-slice z STORES its row sums at rowsumA + z * stride (summed later in a fixed order); 0: atomicAdd into rowsumA
    int vecC;                      // 16-byte accesses to C (and mask) are legal
    int atomic_out;                // 1: atomicAdd into C instead of store
    long long c_split_stride;      // elements between the C slabs of consecutive k-slices (split-K partials in a workspace), else 0
    int tiles_m, tiles_n;
    int debug;                     // tuning aid (env DLRM_GEMM_DEBUG): 1 skip global loads in the k-loop, 2 skip LDS refill + barrier, 4 skip epilogue
    // ReLU sign bits, one bit per output element (see dlrm_relu_bits_bytes in dlrm_hip.h for the layout): written by the
    // forward epilogue (bits_out), consumed by the dgrad epilogue of the NEXT layer instead of its fp32 act' mask (bits_in)
    unsigned* bits_out;
    const unsigned* bits_in;
    long long bits_nblk;           // 64-column blocks per row band = ceil(N / 64)
    // bf16-storage GEMMs (ARITH = 3): a second, bf16 copy of the result (what the next GEMM of the tower reads); C may then be null
    unsigned short* Cb; long long ldcb;
};

template <bool KC>
__device__ __forceinline__ void load_tile(float4 (&r)[4], const float* __restrict__ P, long long ld, int vec,
                                          long long row0, long long rows_max, long long k0, long long k_end,
                                          int tid) {
    // KC : tile = 128 rows (row0..) x 32 k;  thread -> row (tid>>3)+32i, k-quad tid&7
    // !KC: tile = 32 k-rows (k0..) x 128 columns (row0..); thread -> k-row (tid>>5)+8i, col-quad tid&31
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long gr, gc, gr_max, gc_max;
        if (KC) { gr = row0 + (tid >> 3) + 32 * i; gc = k0 + (tid & 7) * 4; gr_max = rows_max; gc_max = k_end; }
        else    { gr = k0 + (tid >> 5) + 8 * i;    gc = row0 + (tid & 31) * 4; gr_max = k_end; gc_max = rows_max; }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < gr_max) {
            const float* p = P + gr * ld + gc;
            if (vec) {
                if (gc < gc_max) v = *(const float4*)p;     // extents are multiples of 4 on this path
            } else {
                if (gc + 0 < gc_max) v.x = p[0];
                if (gc + 1 < gc_max) v.y = p[1];
                if (gc + 2 < gc_max) v.z = p[2];
                if (gc + 3 < gc_max) v.w = p[3];
            }
        }
        r[i] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&r)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* p = KC ? s + ((tid >> 3) + 32 * i) * LD_KC + (tid & 7) * 4
                      : s + ((tid >> 5) + 8 * i) * LD_KS + (tid & 31) * 4;
        *(float4*)__builtin_assume_aligned(p, 16) = r[i];
    }
}

// fragment for the 32-row sub-tile starting at `sub` (0..127), k-group kk (8 k-values):
// lane l supplies row sub + (l&31) and k-values kk*8 + 4*(l>>5) + {0,1,2,3}
template <bool KC>
__device__ __forceinline__ float4 load_frag(const float* __restrict__ s, int sub, int kk, int lane) {
    if (KC) {
        return *(const float4*)__builtin_assume_aligned(s + (sub + (lane & 31)) * LD_KC + kk * 8 + 4 * (lane >> 5), 16);
    } else {
        const float* p = s + (kk * 8 + 4 * (lane >> 5)) * LD_KS + sub + (lane & 31);
        return make_float4(p[0], p[LD_KS], p[2 * LD_KS], p[3 * LD_KS]);
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == DLRM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DLRM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
// derivative of the activation expressed through its OUTPUT y (what the forward pass saved)
__device__ __forceinline__ float act_grad(float g, float y, int act) {
    if (act == DLRM_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DLRM_ACT_SIGMOID) return g * ((1.f - y) * y);
    return g;
}

// staging region of one wave in the epilogue: its 64 (m) x 64 (n) result, rows padded to 68 floats
constexpr int EPI_LD = 64 + 4;
static_assert(4 * 64 * EPI_LD <= 2 * 2 * TILE_F, "epilogue staging does not fit the tile buffers");

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap: hardware places workgroup b on XCD b%8; give each XCD a contiguous range of
    // logical tile ids so the tiles_n workgroups that share one A row panel hit the same L2.
    const int nwg = g.tiles_m * g.tiles_n;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, local = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tile_m = id / g.tiles_n, tile_n = id - tile_m * g.tiles_n;
    const long long m0 = (long long)tile_m * BM, n0 = (long long)tile_n * BN;
    const long long k_begin = (long long)blockIdx.z * g.kchunk;
    const long long k_end = (k_begin + g.kchunk < g.K) ? k_begin + g.kchunk : g.K;
    const int nk = (int)((k_end - k_begin + BK - 1) / BK);

    // acc[tm][tn] holds the TRANSPOSED 32x32 sub-tile (MFMA row index = n, column index = m), so that
    // a lane owns 4 consecutive n of one m per register quad -> 16-byte epilogue traffic.
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // bias gradient of the wgrad GEMM = row sums of the A operand (dZ^T), taken from the fragments the
    // MFMAs consume anyway; only the workgroups of the first n-tile column and their wn == 0 waves add it.
    const bool do_rowsum = g.rowsumA != nullptr && tile_n == 0 && wn == 0;
    float rs[2] = {0.f, 0.f};

    float4 ra[4], rb[4];
    if (nk > 0) {
        load_tile<A_KC>(ra, g.A, g.lda, g.vecA, m0, g.M, k_begin, k_end, tid);
        load_tile<B_KC>(rb, g.B, g.ldb, g.vecB, n0, g.N, k_begin, k_end, tid);
        store_tile<A_KC>(lds, ra, tid);
        store_tile<B_KC>(lds + TILE_F, rb, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const float* sA = lds + (kt & 1) * 2 * TILE_F;
        const float* sB = sA + TILE_F;
        const bool more = kt + 1 < nk;
        if (more && !(g.debug & 1)) {
            const long long k0 = k_begin + (long long)(kt + 1) * BK;
            load_tile<A_KC>(ra, g.A, g.lda, g.vecA, m0, g.M, k0, k_end, tid);
            load_tile<B_KC>(rb, g.B, g.ldb, g.vecB, n0, g.N, k0, k_end, tid);
        }
        // fragments are double buffered in registers: the LDS reads of k-group kk+1 are in flight
        // while the 16 MFMAs of k-group kk issue
        float4 fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[0][t] = load_frag<A_KC>(sA, wm * 64 + t * 32, 0, lane);
            fb[0][t] = load_frag<B_KC>(sB, wn * 64 + t * 32, 0, lane);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[nxt][t] = load_frag<A_KC>(sA, wm * 64 + t * 32, kk + 1, lane);
                    fb[nxt][t] = load_frag<B_KC>(sB, wn * 64 + t * 32, kk + 1, lane);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this k-group's MFMAs
            if (do_rowsum) {
#pragma unroll
                for (int t = 0; t < 2; ++t) rs[t] += (fa[cur][t].x + fa[cur][t].y) + (fa[cur][t].z + fa[cur][t].w);
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][tn].x, fa[cur][tm].x, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][tn].y, fa[cur][tm].y, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][tn].z, fa[cur][tm].z, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][tn].w, fa[cur][tm].w, acc[tm][tn], 0, 0, 0);
                }
        }
        if (more && !(g.debug & 2)) {
            float* dA = lds + ((kt + 1) & 1) * 2 * TILE_F;
            store_tile<A_KC>(dA, ra, tid);
            store_tile<B_KC>(dA + TILE_F, rb, tid);
        }
        if (!(g.debug & 2)) __syncthreads();
    }
    if (g.debug & 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) g.C[0] = s;   // keeps the accumulators live
        return;
    }

    if (do_rowsum) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v = rs[t] + __shfl_xor(rs[t], 32, 64);   // the two half-waves hold different k of the same row
            const long long m = m0 + wm * 64 + t * 32 + (lane & 31);
            if (lane < 32 && m < g.M) {
                if (g.rowsum_split_stride) g.rowsumA[(long long)blockIdx.z * g.rowsum_split_stride + m] = v;
                else atomicAdd(g.rowsumA + m, v);
            }
        }
    }

    // ---- epilogue: registers -> wave-private LDS staging (transposes back to [m][n]) -> 16-byte rows.
    // The k-loop ended with a barrier, so the tile buffers are free.  Transposed C/D layout of the 32x32
    // MFMA: lane owns m_local = lane & 31 and n_local = 8*q + 4*(lane>>5) + {0,1,2,3} for q = reg>>2.
    float* S = lds + wave * (64 * EPI_LD);
    {
        const int ml = lane & 31, h = lane >> 5;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                    *(float4*)__builtin_assume_aligned(S + (tm * 32 + ml) * EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
                }
    }
    // same wave wrote and reads: LDS operations of a wave complete in order
    const int c4 = (lane & 15) * 4;
    const long long nb = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) {
        if (nb + 0 < g.N) bv.x = g.bias[nb + 0];
        if (nb + 1 < g.N) bv.y = g.bias[nb + 1];
        if (nb + 2 < g.N) bv.z = g.bias[nb + 2];
        if (nb + 3 < g.N) bv.w = g.bias[nb + 3];
    }
    const bool full_n = nb + 3 < g.N;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const long long m = m0 + wm * 64 + row;
        if (m >= g.M || nb >= g.N) continue;
        float4 v = *(const float4*)__builtin_assume_aligned(S + row * EPI_LD + c4, 16);
        v.x = act_apply(v.x + bv.x, g.act); v.y = act_apply(v.y + bv.y, g.act);
        v.z = act_apply(v.z + bv.z, g.act); v.w = act_apply(v.w + bv.w, g.act);
        float* c = g.C + (long long)blockIdx.z * g.c_split_stride + m * g.ldc + nb;
        if (g.vecC && full_n) {
            if (g.mask) {
                const float4 y = *(const float4*)(g.mask + m * g.ldmask + nb);
                v.x = act_grad(v.x, y.x, g.mask_act); v.y = act_grad(v.y, y.y, g.mask_act);
                v.z = act_grad(v.z, y.z, g.mask_act); v.w = act_grad(v.w, y.w, g.mask_act);
            }
            if (g.atomic_out) { atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w); }
            else *(float4*)c = v;
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if (nb + x < g.N) {
                    float o = e[x];
                    if (g.mask) o = act_grad(o, g.mask[m * g.ldmask + nb + x], g.mask_act);
                    if (g.atomic_out) atomicAdd(c + x, o); else c[x] = o;
                }
            }
        }
    }
}

// =============================================================================================
// v3: LDS-DMA fed kernel (the fast path; gemm_f32_kernel above stays as the any-shape fallback).
//
// What limits gemm_f32_kernel (measured, profiles/r01): at a 128x128 tile the L2->LDS stream is
// 32 FLOP/B (~7 B/clk/CU at the fp32 MFMA rate, against ~10 B/clk/CU the vector memory path
// sustains), the register-staged refill costs 8 global loads + 8 ds_write_b128 + exec-masked
// bounds code per k-tile, and the single barrier per k-tile sits behind a vmcnt(0).  v3:
//   * tile 256x128x16 (TM=4: four waves of 128x64) or 128x128x16 (TM=2): 43.7 / 32 FLOP per byte;
//   * global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write, no VALU
//     address math: per-lane 32-bit offsets are loop invariant, the k advance is one scalar add
//     on the SGPR base), 3-stage ring, tile kt+2 is issued right after the barrier of tile kt and
//     retired by a COUNTED `s_waitcnt vmcnt(P)` — a prefetch is always in flight across the barrier;
//   * k-contiguous tiles ([rows][16] floats, 64-B rows) are XOR-swizzled through the SOURCE
//     address (the DMA destination is lane-linear): 16-B slot s of row r holds k-quad
//     s ^ ((r>>2)&3), which makes every ds_read_b128 fragment read conflict free;
//     k-strided tiles ([16][cols]) are read with conflict-free ds_read_b32.
// Preconditions (checked by the host dispatch, otherwise the fallback runs): 16-byte aligned
// operands, leading dimensions % 4 == 0, reduction length of every k-slice % 16 == 0.
// Rows/columns past the matrix edge are clamped to a valid address (their results are never
// stored), so M and N are unrestricted.
// =============================================================================================
constexpr int BK3 = 16;

// ---- fp32 through the bf16 matrix pipe (ARITH = 1, "bf16x6") --------------------------------------
// An fp32 value has a 24-bit significand; truncating to bf16 three times,
//     h = trunc_bf16(x),  m = trunc_bf16(x - h),  l = x - h - m      (both subtractions exact in fp32),
// gives three bf16 numbers with x == h + m + l EXACTLY (8 significant bits each; barring underflow below
// 2^-126).  a*b is then the sum of nine exact bf16 products; the six of order >= 2^-16,
//     ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm),
// are issued as six v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped terms (am*bl, al*bm, al*bl) are
// <= 2^-23 |a*b| in total — the size of ONE fp32 rounding of the product — so the result stays in the fp32
// round-off class while the matrix pipe runs at 16/6 = 2.7x the native fp32 MFMA rate.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
struct Split3 { uintx4 h, m, l; };

__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        r.h[i] = __builtin_amdgcn_perm(ub, ua, 0x07060302);            // upper halves = truncation to bf16
        const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
        const unsigned ura = __float_as_uint(ra), urb = __float_as_uint(rb);
        r.m[i] = __builtin_amdgcn_perm(urb, ura, 0x07060302);
        const float sa = ra - __uint_as_float(ura & 0xffff0000u), sb = rb - __uint_as_float(urb & 0xffff0000u);
        r.l[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);   // <= 8 bits left: exact
    }
    return r;
}
// ---- plain bf16 operands (ARITH = 2, "bf16": BASELINE.json configs[4], "bf16 MLP on MFMA") -----------------------
// each fp32 operand is rounded to the nearest bf16 (ties to even, the rounding torch.bfloat16 conversion uses), ONE
// v_mfma_f32_32x32x16_bf16 per 16-k step, fp32 accumulation and fp32 results: 16x the fp32 MFMA rate, ~3 significant
// decimal digits per operand.
// (v_cvt_pk_bf16_f32: ONE instruction converts two fp32 values to a packed bf16 pair with round-to-nearest-even — the manual
// add-0x7fff-and-mask sequence of rounds 1-2 cost ~8 VALU instructions per pair inside the k-loop, which made this path VALU-bound:
// 289 TFLOP/s = 0.12 of the bf16 MFMA peak, profiles/r03/bench_tb_bf16.json)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uintx4 round8_bf16(const float (&x)[8]) {
    uintx4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
    return r;
}
#define MFMA_BF16(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)

__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    // M0 carries the wave-uniform LDS destination; written in the same statement that uses it
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// per-lane byte offset (relative to the tile origin pointer) of DMA chunk `c` (1 KiB of the LDS tile image)
template <bool KC, int ROWS>
__device__ __forceinline__ unsigned dma_offset(int c, int lane, long long ld, long long row0, long long rows_max) {
    if (KC) {        // tile image [ROWS][16]: chunk = 16 rows x 4 slots
        const int r = 16 * c + (lane >> 2);
        const int kq = (lane & 3) ^ ((r >> 2) & 3);
        long long rg = row0 + r; if (rg > rows_max - 1) rg = rows_max - 1;
        return (unsigned)(((rg - row0) * ld + 4 * kq) * 4);
    } else {         // tile image [16][ROWS]: chunk = 256 consecutive floats
        const int e = 256 * c + 4 * lane;
        const int k = e / ROWS, col = e - k * ROWS;
        const long long cg = row0 + col;
        return (unsigned)((k * ld + (cg < rows_max ? col : 0)) * 4);
    }
}

// FRAG = 1 (native fp32 only): VECTOR fragments for k-strided operands.  The LDS image of such an operand is [16 k][cols]; a lane
// feeds sub-tile t of its wave with column (t*32 + r) at four k rows — four ds_read_b32 per sub-tile and k-group, 48 scalar LDS
// reads per 16-k tile in the weight-gradient GEMM where BOTH operands are k-strided.  With FRAG the wave's sub-tiles are
// INTERLEAVED instead of stacked: sub-tile t owns columns TM*r + t (r = 0..31), so the TM (TN) values a lane needs at one k are
// adjacent and ONE ds_read_b128 (b64) at a fixed k row feeds all its sub-tiles: 12 vector reads per 16-k tile instead of 48
// scalar ones (conflict-free: 32 lanes read 512 / 256 contiguous bytes).  The interleave is undone for free by the epilogue,
// which already passes every band through LDS: band t's staged rows are the output rows TM*row + t, and the two column
// sub-tiles are written to LDS column-interleaved so that a staged row is again a contiguous run of output columns.
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int N> struct FVec;
struct FloatX1 { float v; __device__ __forceinline__ float operator[](int) const { return v; } };
template <> struct FVec<1> { using T = FloatX1; };
template <> struct FVec<2> { using T = floatx2; };
template <> struct FVec<4> { using T = floatx4; };

// TN = 1 (with TM = 1: a 64 x 64 tile, four waves of 32 x 32): SMALL launches only (Criteo-Kaggle's batch of 2048, see launch_gemm) — no sign
// bits in or out (the host takes the fp32 mask / the stand-alone bit kernel), the mask of the data gradient is read in the epilogue.
// SCHED: fragment-read schedule of the native fp32 main loop (0 / 1, see the loop); NST: stages of the LDS ring.
// (Round 6 also built and measured, inside one visit each: fragment reads ROLLING across k-tiles — no product ever waits for an LDS read — and the
// accumulators pinned to AGPRs: no gain / +0.5 % forward, -8 % data gradient.  Both were removed again; commit a2c519a has them.)
// EPI: 0 = the general epilogue (any shape / alignment, atomics, fp32 masks, every activation); 1 / 2 = the STRAIGHT-LINE epilogue (no / ReLU activation)
template <bool A_KC, bool B_KC, int TM, int ARITH, bool ROWSUM, int FRAG = 0, int TN = 2, int SCHED = 0, int NST = 3, int EPI = 0>
__global__ __launch_bounds__(256, (NST == 2 && TM <= 2) ? 4 : 2) void gemm3_kernel(GemmArgs g) {
    static_assert(SCHED == 0 || ARITH == 0, "fragment schedules exist for the native fp32 main loop");
    // NST = 3: tile kt + 2 is in flight while tile kt is multiplied (two k-tiles of flight time).  NST = 2 (round 6, 128-row fp32 tiles): one tile of
    // lookahead, 32 KB of ring (34 KB with the epilogue staging) -> FOUR workgroups per CU instead of three (tools/probes/mfma_lds_probe.hip:
    // the MFMA + ds_read stream of this loop sustains 0.958 of the peak at four waves per SIMD, 0.941 at three)
    static_assert(NST == 2 || NST == 3, "two- or three-stage ring");
    static_assert(SCHED == 0 || SCHED == 1, "fragment schedules 0 / 1");
    static_assert(TN == 2 || (TN == 1 && TM == 1 && ARITH == 0), "the 32-column wave tile exists for the small fp32 launches only");
    constexpr bool A_IL = FRAG && !A_KC, B_IL = FRAG && !B_KC;      // operand's sub-tiles interleaved (see above)
    static_assert(!FRAG || ARITH == 0, "vector fragments are implemented for the native fp32 main loop");
    static_assert(ARITH != 3 || (A_KC && B_KC), "bf16-storage operands are k-contiguous (the data gradient reads a transposed weight copy)");
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    constexpr int A_BYTES = BMt * BK3 * 4, B_BYTES = BNt * BK3 * 4, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // Workgroups are handed to the 8 XCDs round-robin in dispatch order (x fastest, then z); each XCD has its own L2.  The dispatch
    // index is re-mapped so that every XCD works on a CONTIGUOUS range of the (k-slice, tile) space:
    //   * one k-slice (forward / data gradient): neighbouring tiles — they share an A row panel — meet in one L2;
    //   * split-k (weight gradient): all output tiles of a k-slice sit on ONE XCD and march through the same rows of dY and X
    //     together, so HBM delivers every row once per launch instead of once per XCD (PMC FETCH_SIZE of the five big layers: 926 -> 348 MB per
    //     launch = 1.00 x the algorithmic dY + X bytes; same step time — the kernel is MFMA-bound — profiles/r03/ceilings.md).
    const int nwg = g.tiles_m * g.tiles_n;
    int id, zs;
    {
        const int total = nwg * (int)gridDim.z, lin = (int)blockIdx.z * nwg + (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, local = lin >> 3;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        zs = w / nwg; id = w - zs * nwg;
        if (g.debug & 32) {                  // tuning aid: the round-3 mapping (tiles re-mapped inside every k-slice on its own)
            const int q1 = nwg >> 3, r1 = nwg & 7, x1 = (int)blockIdx.x & 7, l1 = (int)blockIdx.x >> 3;
            id = (x1 < r1 ? x1 * (q1 + 1) : r1 * (q1 + 1) + (x1 - r1) * q1) + l1; zs = (int)blockIdx.z;
        }
    }
    // (integer division runs on the vector ALU: the quotients are wave-uniform but live in VGPRs.  The DMA base pointers derived
    // from them are "s" operands of the inline asm below — pinned to SGPRs here, not left to the register allocator)
    zs = __builtin_amdgcn_readfirstlane(zs);
    const int tile_m = __builtin_amdgcn_readfirstlane(id / g.tiles_n), tile_n = __builtin_amdgcn_readfirstlane(id - tile_m * g.tiles_n);
    const long long m0 = (long long)tile_m * BMt, n0 = (long long)tile_n * BNt;
    const long long k_begin = (long long)zs * g.kchunk;
    const long long k_end = (k_begin + g.kchunk < g.K) ? k_begin + g.kchunk : g.K;
    const int nk = (int)((k_end - k_begin) / BK3);
    // ---- DMA plan: wave w owns chunks w, w+4, ... of the A image (TM of them) and of the B image (TN)
    unsigned offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = dma_offset<A_KC, BMt>(wave + 4 * i, lane, g.lda, m0, g.M);
#pragma unroll
    for (int i = 0; i < TN; ++i) offB[i] = dma_offset<B_KC, BNt>(wave + 4 * i, lane, g.ldb, n0, g.N);
    const char* baseA = (const char*)(A_KC ? g.A + m0 * g.lda + k_begin : g.A + k_begin * g.lda + m0);
    const char* baseB = (const char*)(B_KC ? g.B + n0 * g.ldb + k_begin : g.B + k_begin * g.ldb + n0);
    const long long stepA = A_KC ? (long long)BK3 * 4 : (long long)BK3 * 4 * g.lda;
    const long long stepB = B_KC ? (long long)BK3 * 4 : (long long)BK3 * 4 * g.ldb;
    const unsigned dstA = lds_base + wave * 1024, dstB = lds_base + A_BYTES + wave * 1024;

#define GEMM3_ISSUE(stage_off)                                                            \
    do {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) glds16(offA[i], baseA, dstA + (stage_off) + i * 4096); \
        _Pragma("unroll") for (int i = 0; i < TN; ++i) glds16(offB[i], baseB, dstB + (stage_off) + i * 4096); \
        baseA += stepA; baseB += stepB;                                                   \
    } while (0)

    // ---- fragment read offsets (bytes inside a stage)
    const int l31 = lane & 31, h = lane >> 5;
    unsigned fa_off[2], fb_off[2];      // j = 0, 1 (the two 8-k groups of a 16-k tile)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        fa_off[j] = A_KC ? (unsigned)((wm * 32 * TM + l31) * 64 + (((2 * j + h) ^ ((l31 >> 2) & 3)) * 16))
                         : (unsigned)((((8 * j + 4 * h) * BMt) + wm * 32 * TM + (A_IL ? TM * l31 : l31)) * 4);
        fb_off[j] = (unsigned)A_BYTES +
                    (B_KC ? (unsigned)((wn * 32 * TN + l31) * 64 + (((2 * j + h) ^ ((l31 >> 2) & 3)) * 16))
                          : (unsigned)((((8 * j + 4 * h) * BNt) + wn * 32 * TN + (B_IL ? TN * l31 : l31)) * 4));
    }

    // bf16x6 operand layout: lane half h owns k = 8h..8h+7 -> k-quads 2h, 2h+1 (k-contiguous) or k-rows 8h.. (k-strided)
    unsigned fas_off[2], fbs_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        fas_off[q] = A_KC ? (unsigned)((wm * 32 * TM + l31) * 64 + (((2 * h + q) ^ ((l31 >> 2) & 3)) * 16))
                          : (unsigned)(((8 * h * BMt) + wm * 32 * TM + l31) * 4);
        fbs_off[q] = (unsigned)A_BYTES +
                     (B_KC ? (unsigned)((wn * 32 * TN + l31) * 64 + (((2 * h + q) ^ ((l31 >> 2) & 3)) * 16))
                           : (unsigned)(((8 * h * BNt) + wn * 32 * TN + l31) * 4));
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_rowsum = ROWSUM && g.rowsumA != nullptr && tile_n == 0 && wn == 0;
    float rs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rs[i] = 0.f;

    // ---- dgrad only: the previous layer's activation (the act' mask of the epilogue) is PREFETCHED.  Every
    // workgroup of a launch reaches its epilogue at about the same time, so a load issued there is pure exposed
    // latency for the matrix cores (measured: 1083 us with the in-epilogue read vs 984 us without a mask on the
    // 1024x1024 layer).  The 8 row segments of the first 32-row band are requested while the last two k-tiles are
    // still being multiplied, band b+1 is requested before band b is transposed and stored.
    constexpr bool MASKED = (A_KC && !B_KC) || ARITH == 3;      // (bf16 storage: forward and data gradient share the <KC, KC> instance)
    const int c4 = TN == 2 ? (lane & 15) * 4 : (lane & 7) * 4;
    const long long nb = n0 + wn * (32 * TN) + c4;
    const bool full_n = nb + 3 < g.N;
    const bool use_bits = TN == 2 && MASKED && g.bits_in != nullptr;
    const bool mask_pf = TN == 2 && MASKED && !use_bits && g.mask != nullptr && g.vecC && full_n;
    float4 mk[2][8];
    // bit form of the same mask: every lane holds ITS 32 sign bits of the band (one dword: 256 B per 32 x 64 band instead of 8 KB)
    float4 bvf = make_float4(0.f, 0.f, 0.f, 0.f);      // EPI != 0: this lane's bias quad (prefetched in the k-loop)
    unsigned mkb[TM];       // ALL bands' words are loaded before the first store of the epilogue: a load issued between the store bursts makes
                            // the compiler drain vmcnt — i.e. wait for every store issued so far to complete — before the word is used
                            // (measured: the masked data-gradient GEMM 4-12 % slower than the unmasked one; with this, equal)
#pragma unroll
    for (int i = 0; i < TM; ++i) mkb[i] = 0u;
    auto bits_fetch = [&](int band, unsigned& dst) {
        const long long mb = (m0 + wm * 32 * TM + band * 32) >> 5;
        const long long nbk = (n0 + wn * 64) >> 6;
        const long long last_band = (g.M - 1) >> 5;
        dst = (nbk < g.bits_nblk) ? g.bits_in[((mb < last_band ? mb : last_band) * g.bits_nblk + nbk) * 64 + lane] : 0u;
    };
    auto mask_fetch = [&](int band, float4 (&dst)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            long long m = m0 + wm * 32 * TM + band * 32 + it * 4 + (lane >> 4);
            if (m > g.M - 1) m = g.M - 1;                       // clamped rows are never stored
            dst[it] = *(const float4*)(g.mask + m * g.ldmask + nb);
        }
    };

    if (nk > 0) GEMM3_ISSUE(0);
    if (nk > 1 && NST > 2) GEMM3_ISSUE(STAGE);
    unsigned cur = 0, nxt = NST > 2 ? 2 * STAGE : STAGE;     // byte offsets of the stage being read / being refilled
    const char* ldsb = (const char*)lds;
    // fragments of one 8-k half (j) of the tile in the stage at byte offset st_off -> fa[TM], fb[TN]
    auto load_half = [&](unsigned st_off, int j, float4 (&fa)[TM], float4 (&fb)[TN]) {
        if constexpr (A_IL) {           // one vector read per k row feeds all TM sub-tiles (rows TM*r + t)
            using VA = typename FVec<TM>::T;
            const char* pa = ldsb + st_off + fa_off[j];
            const VA a0 = *(const VA*)(pa), a1 = *(const VA*)(pa + BMt * 4), a2 = *(const VA*)(pa + 2 * BMt * 4), a3 = *(const VA*)(pa + 3 * BMt * 4);
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = make_float4(a0[t], a1[t], a2[t], a3[t]);
        } else {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (A_KC) fa[t] = *(const float4*)(ldsb + st_off + fa_off[j] + t * 32 * 64);
                else {
                    const float* p = (const float*)(ldsb + st_off + fa_off[j]) + t * 32;
                    fa[t] = make_float4(p[0], p[BMt], p[2 * BMt], p[3 * BMt]);
                }
            }
        }
        if constexpr (B_IL) {
            using VB = typename FVec<TN>::T;
            const char* pb = ldsb + st_off + fb_off[j];
            const VB b0 = *(const VB*)(pb), b1 = *(const VB*)(pb + BNt * 4), b2 = *(const VB*)(pb + 2 * BNt * 4), b3 = *(const VB*)(pb + 3 * BNt * 4);
#pragma unroll
            for (int t = 0; t < TN; ++t) fb[t] = make_float4(b0[t], b1[t], b2[t], b3[t]);
        } else {
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_KC) fb[t] = *(const float4*)(ldsb + st_off + fb_off[j] + t * 32 * 64);
                else {
                    const float* p = (const float*)(ldsb + st_off + fb_off[j]) + t * 32;
                    fb[t] = make_float4(p[0], p[BNt], p[2 * BNt], p[3 * BNt]);
                }
            }
        }
    };
    // the 4 x TM x TN products of one half (k-step major: every accumulator is touched once per k-step)
    auto products = [&](const float4 (&fa)[TM], const float4 (&fb)[TN]) {
        if (do_rowsum) {
#pragma unroll
            for (int t = 0; t < TM; ++t) rs[t] += (fa[t].x + fa[t].y) + (fa[t].z + fa[t].w);
        }
        if constexpr (!A_KC && !B_KC && !FRAG) __builtin_amdgcn_s_setprio(1);    // weight gradient with SCALAR fragments (32 ds_read_b32 per half tile): -1.5 %; nil elsewhere
#define GEMM3_KSTEP(E)                                                                                              \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                           \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[tn].E, fa[tm].E, acc[tm][tn], 0, 0, 0);
        GEMM3_KSTEP(x) GEMM3_KSTEP(y) GEMM3_KSTEP(z) GEMM3_KSTEP(w)
#undef GEMM3_KSTEP
        if constexpr (!A_KC && !B_KC && !FRAG) __builtin_amdgcn_s_setprio(0);
    };
    float4 fa0[TM], fb0[TN], fa1[SCHED ? TM : 1], fb1[SCHED ? TN : 1];      // fragment register sets (native fp32 loop)
    for (int kt = 0; kt < nk; ++kt) {
        if (!(g.debug & 2)) {
            if (NST > 2 && kt + 1 < nk) wait_vmcnt<TM + TN>(); else wait_vmcnt<0>();   // my share of tile kt has landed (NST = 2: it is the only refill in flight)
            __builtin_amdgcn_s_barrier();      // everyone's share has; everyone is done reading stage `nxt`
        }
        // (issued BEFORE this tile's fragment reads: issuing it after them — the reads are on the first MFMA's critical path, the refill is not —
        // measured 1-2 % slower in every kernel form, profiles/round5/gemm_dma_late_ab.txt: the refill's head start matters more)
        if (kt + (NST - 1) < nk && !(g.debug & 1)) GEMM3_ISSUE(nxt);
        if constexpr (EPI != 0) {
            // (straight-line epilogue) the bias quad of this lane's columns, requested with the last two k-tiles still to multiply — as the mask words
            // below: behind every DMA in the in-order vmcnt queue, retired by the loop's final vmcnt(0)
            if (kt + 2 == nk && g.bias != nullptr && nb < g.N) bvf = *(const float4*)(g.bias + nb);
        }
        if constexpr (MASKED) {
            // no DMA is issued after tile nk-1's (at kt == nk-3), so these loads sit behind every DMA in the in-order
            // vmcnt queue and the counted wait of kt == nk-2 (which leaves TM+TN newer operations in flight) and the
            // final vmcnt(0) stay correct
            if (mask_pf && kt + 2 == nk) mask_fetch(0, mk[0]);
            if (use_bits && kt + 2 == nk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) bits_fetch(i, mkb[i]);
            }
        }
        if constexpr (ARITH == 0) {
        // One 8-k half of a tile = TM + TN fragment reads feeding 4 x TM x TN MFMAs.  Two schedules of reads against products:
        //   SCHED 0  one register set: the reads of half j sit right in front of its products (an LDS round trip per half, hidden only by
        //            the other waves of the SIMD);
        //   SCHED 1  (round 5, weight gradient) two sets, both halves' reads in front of the tile's first product: one round trip per tile;
        if constexpr (SCHED == 1) {
            load_half(cur, 0, fa0, fb0);
            load_half(cur, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);      // keep all fragment reads of the tile in front of its first MFMA
            products(fa0, fb0);
            products(fa1, fb1);
        } else {
            load_half(cur, 0, fa0, fb0);
            products(fa0, fb0);
            load_half(cur, 1, fa0, fb0);
            products(fa0, fb0);
        }
        } else if constexpr (ARITH == 3) {
            // bf16 STORAGE: the operands already are bf16 in memory.  A k-contiguous bf16 row tile is, byte for byte, the fp32 tile
            // image with half as many "floats" per row — the same DMA plan, the same XOR swizzle, the same ds_read_b128 — and the
            // 16 bytes a lane reads (fp32 k-quad 2j + h of its row) are exactly the 8 consecutive bf16 k-values
            // 16j + 8h .. + 7 that v_mfma_f32_32x32x16_bf16 wants from it: one MFMA per (sub-tile pair, j), nothing converted in the loop.
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uintx4 fa[TM], fb[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t] = *(const uintx4*)(ldsb + cur + fa_off[j] + t * 32 * 64);
#pragma unroll
                for (int t = 0; t < TN; ++t) fb[t] = *(const uintx4*)(ldsb + cur + fb_off[j] + t * 32 * 64);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = MFMA_BF16(fb[tn], fa[tm], acc[tm][tn]);
            }
        } else {
            // 32x32x16 bf16 operand: lane supplies row (lane & 31), k = 8*(lane>>5) + 0..7 -> the whole 16-k tile is one step
            Split3 sb[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                float x[8];
                if (B_KC) {
                    const float4 v0 = *(const float4*)(ldsb + cur + fbs_off[0] + t * 32 * 64);
                    const float4 v1 = *(const float4*)(ldsb + cur + fbs_off[1] + t * 32 * 64);
                    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
                } else {
                    const float* p = (const float*)(ldsb + cur + fbs_off[0]) + t * 32;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = p[e * BNt];
                }
                if constexpr (ARITH == 1) sb[t] = split3(x); else sb[t].h = round8_bf16(x);
            }
            // one 32-row band of A at a time: read + split its 8 operands, then its 6 x TN products (smallest terms
            // first; the TN accumulators of the band alternate).  Measured alternatives (profiles/r01/bf16x6_variants.md):
            // forcing a 1 MFMA : 4 VALU interleave with sched_group_barrier, and issuing the products product-major over
            // all bands, were both 10-15 % slower than letting the two waves of a SIMD overlap their VALU and MFMA phases.
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                float x[8];
                if (A_KC) {
                    const float4 v0 = *(const float4*)(ldsb + cur + fas_off[0] + tm * 32 * 64);
                    const float4 v1 = *(const float4*)(ldsb + cur + fas_off[1] + tm * 32 * 64);
                    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
                } else {
                    const float* p = (const float*)(ldsb + cur + fas_off[0]) + tm * 32;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = p[e * BMt];
                }
                if (do_rowsum) rs[tm] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
#define GEMM3_PRODUCT(BP, AP) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = MFMA_BF16(sb[tn].BP, sa.AP, acc[tm][tn]);
                if constexpr (ARITH == 1) {
                    const Split3 sa = split3(x);
                    GEMM3_PRODUCT(l, h) GEMM3_PRODUCT(h, l) GEMM3_PRODUCT(m, m)
                    GEMM3_PRODUCT(m, h) GEMM3_PRODUCT(h, m) GEMM3_PRODUCT(h, h)
                } else {
                    Split3 sa;
                    sa.h = round8_bf16(x);
                    GEMM3_PRODUCT(h, h)
                }
#undef GEMM3_PRODUCT
            }
        }
        cur = (cur == (NST - 1) * STAGE) ? 0 : cur + STAGE;
        nxt = (nxt == (NST - 1) * STAGE) ? 0 : nxt + STAGE;
    }
#undef GEMM3_ISSUE
    wait_vmcnt<0>();
    __syncthreads();                       // all fragment reads done: the ring becomes epilogue staging
    if (g.debug & 4) {                     // tuning aid: no epilogue (keeps the accumulators live)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_ += acc[i][j][r];
        if (s_ == 123.456f) g.C[0] = s_;
        return;
    }

    if (do_rowsum) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float v = rs[t] + __shfl_xor(rs[t], 32, 64);
            const long long m = m0 + wm * 32 * TM + (A_IL ? TM * l31 + t : t * 32 + l31);
            if (lane < 32 && m < g.M) {
                if (g.rowsum_split_stride) g.rowsumA[(long long)zs * g.rowsum_split_stride + m] = v;
                else atomicAdd(g.rowsumA + m, v);
            }
        }
    }

    // ---- epilogue, one 32-row band of the wave tile at a time: registers -> wave-private LDS
    // (transposes back to [m][n]) -> 16-byte row segments.  Transposed C/D layout of the 32x32 MFMA:
    // lane owns m_local = lane & 31 and n_local = 8*q + 4*(lane>>5) + {0..3} for q = reg>>2.
    constexpr int ELD = TN == 2 ? EPI_LD : 32 + 4;                  // staged row pitch (floats)
    // (the staging area of the four waves may be larger than a two-stage ring: launch_gemm3 sizes the dynamic LDS as the maximum of the two)
    if constexpr (MASKED) {
        // the words were requested two k-tiles ago and the loop's final vmcnt(0) has retired them; the compiler cannot see that
        // through the hand-placed waits, and would drain vmcnt (= wait for the previous band's STORES) in front of every later use
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(mkb[i]));
    }
    float* S = lds + wave * (32 * ELD);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) {
        if (nb + 0 < g.N) bv.x = g.bias[nb + 0];
        if (nb + 1 < g.N) bv.y = g.bias[nb + 1];
        if (nb + 2 < g.N) bv.z = g.bias[nb + 2];
        if (nb + 3 < g.N) bv.w = g.bias[nb + 3];
    }
    if constexpr (MASKED) { if (mask_pf && nk < 2) mask_fetch(0, mk[0]); if (use_bits && nk < 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) bits_fetch(i, mkb[i]);
        } }
    const bool write_bits = TN == 2 && A_KC && B_KC && g.bits_out != nullptr;
    if constexpr (EPI != 0) {
        // ---- STRAIGHT-LINE epilogue (round 6).  The general epilogue below carries every option of the entry points in one body (atomic
        // accumulation, scalar edges, fp32 masks read in place, three activations behind run-time branches); the waitcnt pass then closes every
        // join of those branches with s_waitcnt vmcnt(0) — IN FRONT OF EVERY ROW STORE AND EVERY STAGED READ — so a wave's 8 x TM row stores left one
        // full memory round trip apart (disassembly: profiles/round6/gemm3_epilogue_isa.md; timing: the epilogue was 4.7 % of the 1024 x 1024
        // layers and 11-16 % of the 512 -> 256 ones, of which the store traffic itself is a quarter).  This body is what the host selects when
        // the call needs none of that (launch_gemm: 16-byte aligned C rows, N % 4 == 0, plain stores, no fp32 mask, no sigmoid): no loads, no
        // divergent paths — the only VMEM operations are the row stores, and nothing waits for them.  Same arithmetic, same bits.
        static_assert(TN == 2 && ARITH == 0, "the straight-line epilogue exists for the native fp32 128-column tiles");
        const bool col_ok = nb < g.N;                       // (N % 4 == 0: a lane's four columns are all inside or all outside)
        if (nk < 2 && g.bias != nullptr && col_ok) bvf = *(const float4*)(g.bias + nb);
        asm volatile("" : "+v"(bvf.x), "+v"(bvf.y), "+v"(bvf.z), "+v"(bvf.w));      // (retired by the final vmcnt(0) / loaded just above: no wait at its uses)
        float* cbase = g.C + (long long)zs * g.c_split_stride + nb;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            unsigned myword = 0u;
            if constexpr (B_IL) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v0 = make_float4(acc[tm][0][4 * q], acc[tm][1][4 * q], acc[tm][0][4 * q + 1], acc[tm][1][4 * q + 1]);
                    const float4 v1 = make_float4(acc[tm][0][4 * q + 2], acc[tm][1][4 * q + 2], acc[tm][0][4 * q + 3], acc[tm][1][4 * q + 3]);
                    *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + 16 * q + 8 * h, 16) = v0;
                    *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + 16 * q + 8 * h + 4, 16) = v1;
                }
            } else {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                        *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
                    }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const long long m = m0 + wm * 32 * TM + (A_IL ? TM * row + tm : tm * 32 + row);
                float4 v = *(const float4*)__builtin_assume_aligned(S + row * EPI_LD + c4, 16);
                v.x += bvf.x; v.y += bvf.y; v.z += bvf.z; v.w += bvf.w;
                if constexpr (EPI == 2) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                }
                if constexpr (A_KC && B_KC) {
                    if (write_bits) {
                        asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                     "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                     "v_cmp_lt_f32 vcc, 0, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                     "v_cmp_lt_f32 vcc, 0, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                     : "+v"(myword) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "vcc");
                    }
                }
                if constexpr (MASKED) {
                    if (use_bits) {
                        const unsigned wv = mkb[tm];
                        if (!((wv >> (31 - (it * 4 + 0))) & 1u)) v.x = 0.f;
                        if (!((wv >> (31 - (it * 4 + 1))) & 1u)) v.y = 0.f;
                        if (!((wv >> (31 - (it * 4 + 2))) & 1u)) v.z = 0.f;
                        if (!((wv >> (31 - (it * 4 + 3))) & 1u)) v.w = 0.f;
                    }
                }
                if (m < g.M && col_ok) *(float4*)(cbase + m * g.ldc) = v;
            }
            if constexpr (A_KC && B_KC) {
                if (write_bits) {
                    const long long mb = (m0 + wm * 32 * TM + tm * 32) >> 5, nbk = (n0 + wn * 64) >> 6;
                    if (mb <= ((g.M - 1) >> 5) && nbk < g.bits_nblk) g.bits_out[(mb * g.bits_nblk + nbk) * 64 + lane] = myword;
                }
            }
        }
        return;
    }
    if constexpr (TN == 1) {
        // 32-column wave tile: a lane owns 4 columns of rows it*8 + (lane >> 3); the mask (data gradient) is read here — a small launch has no
        // synchronised epilogue burst to hide it from
        const bool mvec = g.mask != nullptr && g.vecC && full_n && (((size_t)g.mask) & 15) == 0 && (g.ldmask & 3) == 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = make_float4(acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]);
            *(float4*)__builtin_assume_aligned(S + l31 * ELD + 8 * q + 4 * h, 16) = v;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const long long m = m0 + wm * 32 + row;                 // (TM == 1: interleaved and stacked row orders coincide)
            if (!(m < g.M && nb < g.N)) continue;
            float4 v = *(const float4*)__builtin_assume_aligned(S + row * ELD + c4, 16);
            v.x = act_apply(v.x + bv.x, g.act); v.y = act_apply(v.y + bv.y, g.act);
            v.z = act_apply(v.z + bv.z, g.act); v.w = act_apply(v.w + bv.w, g.act);
            float* c = g.C + (long long)zs * g.c_split_stride + m * g.ldc + nb;
            if (g.vecC && full_n) {
                if (MASKED && g.mask) {
                    float4 y;
                    if (mvec) y = *(const float4*)(g.mask + m * g.ldmask + nb);
                    else { const float* yp = g.mask + m * g.ldmask + nb; y = make_float4(yp[0], yp[1], yp[2], yp[3]); }
                    v.x = act_grad(v.x, y.x, g.mask_act); v.y = act_grad(v.y, y.y, g.mask_act);
                    v.z = act_grad(v.z, y.z, g.mask_act); v.w = act_grad(v.w, y.w, g.mask_act);
                }
                if (g.atomic_out) { atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w); }
                else *(float4*)c = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (nb + x < g.N) {
                        float o = e[x];
                        if (MASKED && g.mask) o = act_grad(o, g.mask[m * g.ldmask + nb + x], g.mask_act);
                        if (g.atomic_out) atomicAdd(c + x, o); else c[x] = o;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        if constexpr (MASKED) {
            if (mask_pf && tm + 1 < TM) mask_fetch(tm + 1, mk[(tm + 1) & 1]);
        }
        unsigned myword = 0u;                      // forward: this lane's 32 sign bits of the band, shifted in one element at a time
        if constexpr (B_IL && TN == 2) {
            // column sub-tiles interleaved (output column 2*n_local + tn): element e of quad q lands at staged column 16q + 8h + 2e + tn
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v0 = make_float4(acc[tm][0][4 * q], acc[tm][1][4 * q], acc[tm][0][4 * q + 1], acc[tm][1][4 * q + 1]);
                const float4 v1 = make_float4(acc[tm][0][4 * q + 2], acc[tm][1][4 * q + 2], acc[tm][0][4 * q + 3], acc[tm][1][4 * q + 3]);
                *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + 16 * q + 8 * h, 16) = v0;
                *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + 16 * q + 8 * h + 4, 16) = v1;
            }
        } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                *(float4*)__builtin_assume_aligned(S + l31 * EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + (lane >> 4);
            const long long m = m0 + wm * 32 * TM + (A_IL ? TM * row + tm : tm * 32 + row);   // (row sub-tiles interleaved: staged row r of band tm is output row TM*r + tm)
            const bool live = m < g.M && nb < g.N;
            float4 v = *(const float4*)__builtin_assume_aligned(S + row * EPI_LD + c4, 16);
            v.x = act_apply(v.x + bv.x, g.act); v.y = act_apply(v.y + bv.y, g.act);
            v.z = act_apply(v.z + bv.z, g.act); v.w = act_apply(v.w + bv.w, g.act);
            if constexpr (A_KC && B_KC) {
                if (write_bits) {       // word = 2*word + (v > 0): compare into VCC, add-with-carry — two VALU instructions per element
                    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                 : "+v"(myword) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "vcc");
                }
            }
            if (!live) continue;
            float* c = g.C + (long long)zs * g.c_split_stride + m * g.ldc + nb;
            if constexpr (MASKED) {
                if (use_bits) {         // ReLU derivative from the sign bits the forward pass stored: element it*4 + c sits at bit 31 - (it*4 + c)
                    const unsigned wv = mkb[tm];
                    if (!((wv >> (31 - (it * 4 + 0))) & 1u)) v.x = 0.f;
                    if (!((wv >> (31 - (it * 4 + 1))) & 1u)) v.y = 0.f;
                    if (!((wv >> (31 - (it * 4 + 2))) & 1u)) v.z = 0.f;
                    if (!((wv >> (31 - (it * 4 + 3))) & 1u)) v.w = 0.f;
                }
            }
            if constexpr (ARITH == 3) {
                // bf16 storage (host guarantees N % 4 == 0 and aligned outputs): the fp32 result for the consumers that need it (weight
                // gradient, matrix-vector layer, interaction) and/or its bf16 rounding for the next GEMM of the tower
                if (g.C) *(float4*)c = v;
                if (g.Cb) {
                    uint2 pk; pk.x = cvt_pk_bf16(v.x, v.y); pk.y = cvt_pk_bf16(v.z, v.w);
                    *(uint2*)(g.Cb + m * g.ldcb + nb) = pk;
                }
            } else
            if (g.vecC && full_n) {
                if constexpr (MASKED) {
                    if (g.mask && !use_bits) {
                        const float4 y = mk[tm & 1][it];
                        v.x = act_grad(v.x, y.x, g.mask_act); v.y = act_grad(v.y, y.y, g.mask_act);
                        v.z = act_grad(v.z, y.z, g.mask_act); v.w = act_grad(v.w, y.w, g.mask_act);
                    }
                }
                if (g.atomic_out) { atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w); }
                else *(float4*)c = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (nb + x < g.N) {
                        float o = e[x];
                        if (g.mask && !use_bits) o = act_grad(o, g.mask[m * g.ldmask + nb + x], g.mask_act);
                        if (g.atomic_out) atomicAdd(c + x, o); else c[x] = o;
                    }
                }
            }
        }
        if constexpr (A_KC && B_KC) {
            if (write_bits) {
                const long long mb = (m0 + wm * 32 * TM + tm * 32) >> 5, nbk = (n0 + wn * 64) >> 6;
                if (mb <= ((g.M - 1) >> 5) && nbk < g.bits_nblk) g.bits_out[(mb * g.bits_nblk + nbk) * 64 + lane] = myword;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// split-K reduction: dW[n,k] (+)= sum_s part[s][n][k]   (fixed summation order: deterministic)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int N, int K, int splits, const float* __restrict__ part,
                                                            long long ldp, long long slab, float* __restrict__ dW,
                                                            long long lddw, int accumulate, const float* __restrict__ rs_part,
                                                            float* __restrict__ dbias) {
    // bias gradient: the k-slices' row sums of dY^T, summed in slice order (deterministic; no zero-fill, no atomics)
    if (rs_part) {          // one thread per bias element, spread over the grid; slice order fixed, loads unrolled (independent addresses)
        for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long long)gridDim.x * 256) {
            float a = accumulate ? dbias[n] : 0.f;
#pragma unroll 8
            for (int s_ = 0; s_ < splits; ++s_) a += rs_part[(long long)s_ * N + n];
            dbias[n] = a;
        }
    }
    const int kq = (K + VEC - 1) / VEC;
    const long long total = (long long)N * kq;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long n = e / kq;
        const int k = (int)(e - n * kq) * VEC;
        const float* p = part + n * ldp + k;
        float* o = dW + n * lddw + k;
        if (VEC == 4) {
            float4 a = accumulate ? *(const float4*)o : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int s_ = 0; s_ < splits; ++s_) {
                const float4 v = *(const float4*)(p + (long long)s_ * slab);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *(float4*)o = a;
        } else {
            float a = accumulate ? *o : 0.f;
            for (int s_ = 0; s_ < splits; ++s_) a += p[(long long)s_ * slab];
            *o = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dZ = dY ⊙ act'(Y), dbias += colsum(dZ)   (tower outputs whose dY does not come from a dgrad GEMM)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(long long M, int N, const float* __restrict__ dY, long long lddy,
                                                      const float* __restrict__ Y, long long ldy, int act,
                                                      float* __restrict__ dZ, long long lddz, float* __restrict__ dbias,
                                                      int tx, int rows_per_block) {
    __shared__ float red[256];
    const int ty_n = 256 / tx;
    const int cx = threadIdx.x % tx, ry = threadIdx.x / tx;
    const long long n = (long long)blockIdx.y * tx + cx;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = (m_begin + rows_per_block < M) ? m_begin + rows_per_block : M;
    float cs = 0.f;
    if (n < N) {
        for (long long m = m_begin + ry; m < m_end; m += ty_n) {
            const float v = act_grad(dY[m * lddy + n], Y[m * ldy + n], act);
            dZ[m * lddz + n] = v;
            cs += v;
        }
    }
    if (dbias) {
        red[threadIdx.x] = cs;
        __syncthreads();
        for (int s = ty_n >> 1; s > 0; s >>= 1) {
            if (ry < s) red[threadIdx.x] += red[threadIdx.x + s * tx];
            __syncthreads();
        }
        if (ry == 0 && n < N) atomicAdd(dbias + n, red[cx]);
    }
}

// fallback producer of the ReLU sign-bit blocks (shapes the LDS-DMA kernel does not take): one thread per 64-bit word
__global__ __launch_bounds__(256) void relu_bits_kernel(long long M, int N, const float* __restrict__ Y, long long ldy,
                                                        unsigned* __restrict__ bits, long long nblk, long long words) {
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < words; w += (long long)gridDim.x * 256) {
        const int l = (int)(w & 63);
        const long long blk = w >> 6, mb = blk / nblk, nbk = blk - mb * nblk;
        unsigned v = 0u;
        for (int e = 0; e < 32; ++e) {       // element e = it*4 + c of lane l
            const long long m = mb * 32 + (e >> 2) * 4 + (l >> 4), n = nbk * 64 + (l & 15) * 4 + (e & 3);
            v = (v << 1) | ((m < M && n < N && Y[m * ldy + n] > 0.f) ? 1u : 0u);
        }
        bits[w] = v;
    }
}

// fp32 -> bf16 copies for the bf16-storage tower: dst[m, n] = bf16(src[m, n]) (round to nearest even), zero for N <= n < Npad
__global__ __launch_bounds__(256) void cast_bf16_kernel(long long M, int N, int Npad, const float* __restrict__ src, long long lds_,
                                                        unsigned short* __restrict__ dst, long long ldd) {
    const int np2 = Npad / 2;
    const long long total = M * np2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / np2;
        const int n = (int)(e - m * np2) * 2;
        const float a = n < N ? src[m * lds_ + n] : 0.f, b = n + 1 < N ? src[m * lds_ + n + 1] : 0.f;
        *(unsigned*)(dst + m * ldd + n) = cvt_pk_bf16(a, b);
    }
}
// the same for 16-byte aligned rows and N % 8 == 0: a thread converts 8 consecutive values (two 16-byte loads, one 16-byte store), one group per thread
// over a full grid — the [65536, 3456] feature buffer of the DCN-v2 cross network and the [65536, 480] interaction output are cast every step
__global__ __launch_bounds__(256) void cast_bf16_vec8_kernel(long long M, int N, int Npad, const float* __restrict__ src, long long lds_,
                                                             unsigned short* __restrict__ dst, long long ldd) {
    const int np8 = Npad / 8;
    const long long total = M * np8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / np8;
        const int n = (int)(e - m * np8) * 8;
        uintx4 pk = {0u, 0u, 0u, 0u};
        if (n < N) {                                                   // (N % 8 == 0: a group lies entirely inside the row or entirely in the padding)
            const float4 a = *(const float4*)(src + m * lds_ + n), b = *(const float4*)(src + m * lds_ + n + 4);
            pk[0] = cvt_pk_bf16(a.x, a.y); pk[1] = cvt_pk_bf16(a.z, a.w); pk[2] = cvt_pk_bf16(b.x, b.y); pk[3] = cvt_pk_bf16(b.z, b.w);
        }
        *(uintx4*)(dst + m * ldd + n) = pk;
    }
}
// transposed copy (weights: dstT[c, r] = bf16(src[r, c]), zero for R <= r < Rpad): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(int R, int C, int Rpad, const float* __restrict__ src, long long lds_,
                                                          unsigned short* __restrict__ dstT, long long ldd) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(long long)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rpad) dstT[(long long)c * ldd + r] = (unsigned short)(cvt_pk_bf16(tile[tx][i], 0.f) & 0xffffu);
    }
}

// ALL weight copies of a bf16 tower in ONE launch (round 5): tensor i = src [R, C] fp32 -> dst [R, Cpad] bf16 (zero columns C..Cpad-1; nullable)
// and / or dstT [C, Rpad] bf16 = its transpose (zero columns R..Rpad-1; nullable).  A workgroup = one 32 x 32 source tile, read once.  The
// per-layer launches it replaces were 15 of the 70 kernels of the Terabyte bf16 step at ~5 us each (profiles/round5/step_trace_tb_bf16.txt).
struct CastMultiArgs {
    const float* src[DLRM_CAST_MULTI_MAX]; long long lds[DLRM_CAST_MULTI_MAX];
    unsigned short* dst[DLRM_CAST_MULTI_MAX]; long long ldd[DLRM_CAST_MULTI_MAX];
    unsigned short* dstT[DLRM_CAST_MULTI_MAX]; long long lddT[DLRM_CAST_MULTI_MAX];
    int R[DLRM_CAST_MULTI_MAX], C[DLRM_CAST_MULTI_MAX], Cpad[DLRM_CAST_MULTI_MAX], Rpad[DLRM_CAST_MULTI_MAX];
    int tiles_c[DLRM_CAST_MULTI_MAX], tile_start[DLRM_CAST_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(CastMultiArgs a) {
    __shared__ float tile[32][33];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.tile_start[i + 1]) ++i;
    const int t = (int)blockIdx.x - a.tile_start[i];
    const int r0 = (t / a.tiles_c[i]) * 32, c0 = (t % a.tiles_c[i]) * 32;
    const int R = a.R[i], C = a.C[i];
    const float* __restrict__ src = a.src[i];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? src[(long long)r * a.lds[i] + c] : 0.f;
    }
    __syncthreads();
    if (a.dst[i]) {
        unsigned short* __restrict__ d = a.dst[i];
        for (int k = ty; k < 32; k += 8) {
            const int r = r0 + k, c = c0 + tx;
            if (r < R && c < a.Cpad[i]) d[(long long)r * a.ldd[i] + c] = (unsigned short)(cvt_pk_bf16(tile[k][tx], 0.f) & 0xffffu);
        }
    }
    if (a.dstT[i]) {
        unsigned short* __restrict__ d = a.dstT[i];
        for (int k = ty; k < 32; k += 8) {
            const int c = c0 + k, r = r0 + tx;
            if (c < C && r < a.Rpad[i]) d[(long long)c * a.lddT[i] + r] = (unsigned short)(cvt_pk_bf16(tile[tx][k], 0.f) & 0xffffu);
        }
    }
}

// ---- fp32 -> three bf16 planes (arith "bf16x6" with PRE-SPLIT operands: gemm_bf16.hip PL = 3).  The truncation split of split3 above, done ONCE
// per tensor instead of in every k-loop that reads it: x == h + m + l exactly.  dst = planes h, m, l of [M, ldd], `plane` elements apart.
__device__ __forceinline__ void split2_planes(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ph = __builtin_amdgcn_perm(ub, ua, 0x07060302);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned ura = __float_as_uint(ra), urb = __float_as_uint(rb);
    pm = __builtin_amdgcn_perm(urb, ura, 0x07060302);
    const float sa = ra - __uint_as_float(ura & 0xffff0000u), sb = rb - __uint_as_float(urb & 0xffff0000u);
    pl = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);
}
// one thread = 8 consecutive values of a row (Npad % 8 == 0; columns N..Npad-1 are zero); VEC: two 16-byte loads (N % 8 == 0, aligned rows)
template <bool VEC>
__global__ __launch_bounds__(256) void split_bf16x3_kernel(long long M, int N, int Npad, const float* __restrict__ src, long long lds_,
                                                           unsigned short* __restrict__ dst, long long ldd, long long plane) {
    const int np8 = Npad / 8;
    const long long total = M * np8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / np8;
        const int n = (int)(e - m * np8) * 8;
        float x[8];
        if (VEC) {
            if (n < N) {
                const float4 a = *(const float4*)(src + m * lds_ + n), b = *(const float4*)(src + m * lds_ + n + 4);
                x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (n + i < N) ? src[m * lds_ + n + i] : 0.f;
        }
        uintx4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) { unsigned h_, m_, l_; split2_planes(x[2 * i], x[2 * i + 1], h_, m_, l_); ph[i] = h_; pm[i] = m_; pl[i] = l_; }
        unsigned short* d = dst + m * ldd + n;
        *(uintx4*)d = ph; *(uintx4*)(d + plane) = pm; *(uintx4*)(d + 2 * plane) = pl;
    }
}
// transposed (weights: planes of dstT[c, r] = src[r, c], zero for R <= r < Rpad): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void split_bf16x3_t_kernel(int R, int C, int Rpad, const float* __restrict__ src, long long lds_,
                                                             unsigned short* __restrict__ dstT, long long ldd, long long plane) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(long long)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rpad) {
            unsigned ph, pm, pl;
            { unsigned h_, m_, l_; split2_planes(tile[tx][i], 0.f, h_, m_, l_); ph = h_; pm = m_; pl = l_; }
            unsigned short* d = dstT + (long long)c * ldd + r;
            d[0] = (unsigned short)(ph & 0xffffu); d[plane] = (unsigned short)(pm & 0xffffu); d[2 * plane] = (unsigned short)(pl & 0xffffu);
        }
    }
}

static int pow2ceil_i(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <bool A_KC, bool B_KC, int TM, int ARITH, int FRAG = 0, int TN = 2, int SCHED = 0, int EPI = 0, int NST = 3>
static int launch_gemm3(GemmArgs& g, int splits, hipStream_t st) {
    constexpr bool ROWSUM = !A_KC && !B_KC;          // only the weight-gradient GEMM carries the bias-gradient row sums
    constexpr int BMt = 64 * TM, BNt = 64 * TN;
    if (TN == 1) { g.bits_in = nullptr; g.bits_out = nullptr; }       // (the caller falls back to the fp32 mask / the stand-alone bit kernel)
    g.tiles_m = (int)((g.M + BMt - 1) / BMt);
    g.tiles_n = (int)((g.N + BNt - 1) / BNt);
    {   // tuning aid (env DLRM_GEMM_DEBUG): 1 no DMA refill in the k-loop, 2 no wait + barrier, 4 no epilogue — WRONG results, timing only
        static int dbg = -1;
        if (dbg < 0) dbg = DLRM_DEBUG_ENV("DLRM_GEMM_DEBUG", 0x7fffffff);
        g.debug = dbg;
    }
    size_t lds = (size_t)NST * (BMt + BNt) * BK3 * 4;           // 72 KiB (TM=4) / 48 KiB (TM=2) with three stages
    const size_t staging = (size_t)4 * 32 * (TN == 2 ? EPI_LD : 32 + 4) * 4;      // the epilogue's four wave-private staging areas reuse the ring
    if (lds < staging) lds = staging;
    {   // tuning aid (env DLRM_GEMM_LDS_PAD, bytes): extra dynamic LDS per workgroup = fewer resident workgroups per CU (occupancy probe)
        static const int pad = DLRM_TUNE_ENV("DLRM_GEMM_LDS_PAD", 0);
        lds += (size_t)pad;
    }
    static bool attr_done[DLRM_MAX_DEVICES] = {};      // the attribute is per (function, device)
    const int dev = dlrm_current_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm3_kernel<A_KC, B_KC, TM, ARITH, ROWSUM, FRAG, TN, SCHED, NST, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[dev] = true;
    }
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits), block(256);
    hipLaunchKernelGGL((gemm3_kernel<A_KC, B_KC, TM, ARITH, ROWSUM, FRAG, TN, SCHED, NST, EPI>), grid, block, lds, st, g);
    DLRM_LAUNCH_CHECK();
    return 0;
}

static int gemm_path() {   // tuning builds, env DLRM_GEMM_PATH: 0 = auto (default), 2 = force the register-staged fallback kernel
    static const int v = DLRM_TUNE_ENV("DLRM_GEMM_PATH", 0);
    return v;
}

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs& g, int splits, hipStream_t st, int arith, bool* fast = nullptr) {
    if (fast) *fast = false;
    // fast path preconditions: 16-byte vector access to both operands, every k-slice a multiple of 16
    const bool k16 = (g.K % BK3 == 0) && (g.kchunk % BK3 == 0);
    if (gemm_path() != 2 && g.vecA && g.vecB && k16 && g.lda % 4 == 0 && g.ldb % 4 == 0) {
        // Tile height.  Split-k (weight gradient): 256-row tiles when they still give every CU two workgroups, else 128-row tiles.
        // One k-slice (forward, data gradient): 128-row tiles — three workgroups per CU (48 KB of LDS each) — except when the 256-row
        // tiling fits exactly one round of 2 x 256 resident workgroups.  Warm microbench, M = 65536 (profiles/r03/ceilings.md):
        // forward 13 -> 512: 44 -> 36 us, 480 -> 1024: 503 -> 486, 1024 -> 1024: 996 -> 985; data gradient with out = 1024 / K = 512:
        // 547 -> 521, out = 512 / K = 256: 162 -> 146; the one-round shapes (512 -> 256 forward, out = 256 data gradient) lose 1-4 %
        // with the small tiles and keep the large ones; the weight gradient loses 20 % with them.
        const long long wg256 = ((g.M + 255) / 256) * ((g.N + 127) / 128) * splits;
        static const int force_tm = DLRM_TUNE_ENV("DLRM_GEMM_TM", 0);      // tuning aid: 2 forces the 128-row tiles, 4 the round-2 rule
        bool big = g.M >= 256 && wg256 >= 512;
        // (fp32 MFMA only: the bf16x6 / bf16 main loops split or round every fragment they load, and the smaller tiles reuse a fragment
        // for half as many products — bf16x6 step 6.51 -> 6.93 ms with them)
        // Round 6: with the two-stage ring four 128-row workgroups fit a CU, and the 128-row tiles then win or tie on EVERY one-slice shape — also the
        // one-round ones (512 -> 256 forward 130.8 -> 127.7 us, 256 -> 128 data gradient 44.2 -> 40.6) and the long reductions that took the
        // 256-row tiles for a while this round (profiles/round6/gemm_tile_rule_ab.md).  The weight gradient keeps the 256-row tiles.
        if (splits == 1 && force_tm != 4 && arith == DLRM_ARITH_F32) big = false;
        if (force_tm == 2) big = false;
        // k-strided operands (data gradient: W; weight gradient: dY and X) are read with vector fragments over interleaved sub-tiles
        // (FRAG, see gemm3_kernel).  Tuning aids: DLRM_GEMM_FRAG=0 -> the scalar-fragment kernels of rounds 1-2;
        // DLRM_WGRAD_TM=2 -> 128-row weight-gradient tiles (three workgroups per CU; lost 20 % with scalar fragments).
        static const int frag = DLRM_TUNE_ENV("DLRM_GEMM_FRAG", 1), wgrad_tm = DLRM_TUNE_ENV("DLRM_WGRAD_TM", 0);
        if (splits > 1 && wgrad_tm == 2) big = false;
        if (fast) *fast = true;
        // SMALL launches (Criteo-Kaggle: batch 2048): a 128 x 128 tiling leaves most CUs idle and every wave walks its k-loop alone at one
        // SIMD's MFMA rate (29 / 25 / 44 us per forward / data- / weight-gradient GEMM of ~0.5 GFLOP, profiles/round4/kaggle_kernels.md).
        // 64-row tiles (TM = 1) double the workgroups and halve each wave's MFMA chain.  fp32 MFMA only.
        static const int small_tm = DLRM_TUNE_ENV("DLRM_GEMM_SMALL", 2);       // 0 off (keeps the 128-row tiles), 1 64-row tiles only, 2 (default) also 64 x 64
        const long long wg128 = ((g.M + 127) / 128) * ((g.N + 127) / 128) * splits;
        const bool small = small_tm && !big && arith == DLRM_ARITH_F32 && wg128 < 128 && force_tm == 0;
        // ... and 64 x 64 tiles (TN = 1 too) while even the 64 x 128 tiling has fewer workgroups than 3/4 of the CUs; that kernel neither writes
        // nor reads sign bits: `fast` stays false so that dlrm_linear_fwd runs the stand-alone bit kernel, the data gradient takes its fp32 mask
        const long long wg64 = ((g.M + 63) / 64) * ((g.N + 127) / 128) * splits;
        const bool tiny = small && small_tm >= 2 && wg64 < 192 && (g.bits_in == nullptr || g.mask != nullptr);
        if (tiny && fast) *fast = false;
        // fragment-read schedule (gemm3_kernel SCHED) of the native fp32 loop per kernel form: compile-time (-DDLRM_SCHED_FWD=.. etc., tools/build_variant_lib.sh)
        constexpr int FORM = (A_KC && B_KC) ? 0 : (A_KC ? 1 : 2);
        constexpr int SCHED_DEFAULT[3] = {DLRM_SCHED_FWD, DLRM_SCHED_DGRAD, DLRM_SCHED_WGRAD};
        constexpr int SD = SCHED_DEFAULT[FORM];
        // the straight-line epilogue (gemm3_kernel EPI) wherever the call needs nothing else: 16-byte aligned C rows, N % 4 == 0, plain stores, the
        // ReLU derivative from sign bits (or none), bias 16-byte aligned, no sigmoid, fp32 results only.  Tuning builds: DLRM_GEMM_EPI=0 keeps the general one.
        static const int epi_on = DLRM_TUNE_ENV("DLRM_GEMM_EPI", 1);
        const bool fast_epi = epi_on && arith == DLRM_ARITH_F32 && g.vecC && g.N % 4 == 0 && g.ldc % 4 == 0 && !g.atomic_out && g.Cb == nullptr &&
                              (g.mask == nullptr || g.bits_in != nullptr) && (g.bias == nullptr || dlrm_aligned16(g.bias)) &&
                              (g.act == DLRM_ACT_NONE || (g.act == DLRM_ACT_RELU && FORM == 0)) && (g.c_split_stride % 4 == 0);
        // ring depth of the 128-row tiles (gemm3_kernel NST): 2 = four workgroups per CU.  Compile-time per form (-DDLRM_NST_FWD=.. etc.);
        // tuning builds: DLRM_GEMM_NST = three digits (forward, data gradient, weight gradient), each 2 / 3
        constexpr int NST_DEFAULT[3] = {DLRM_NST_FWD, DLRM_NST_DGRAD, DLRM_NST_WGRAD};
#ifdef DLRM_TUNING
        static const int nst_all = DLRM_TUNE_ENV("DLRM_GEMM_NST", -1);
        const int nst = nst_all < 0 ? NST_DEFAULT[FORM] : (FORM == 0 ? nst_all / 100 : FORM == 1 ? (nst_all / 10) % 10 : nst_all % 10);
#define GEMM3_LAUNCH2(FR, EP) (nst == 2 ? launch_gemm3<A_KC, B_KC, 2, 0, FR, 2, SD, EP, 2>(g, splits, st) : launch_gemm3<A_KC, B_KC, 2, 0, FR, 2, SD, EP, 3>(g, splits, st))
#else
#define GEMM3_LAUNCH2(FR, EP) launch_gemm3<A_KC, B_KC, 2, 0, FR, 2, SD, EP, NST_DEFAULT[FORM]>(g, splits, st)
#endif
#define GEMM3_LAUNCH4(FR, EP) launch_gemm3<A_KC, B_KC, 4, 0, FR, 2, SD, EP>(g, splits, st)
#define GEMM3_BIG_OR_NOT(FR)                                                                                   \
        if (fast_epi) {                                                                                        \
            if constexpr (FORM == 0) { if (g.act == DLRM_ACT_RELU) return big ? GEMM3_LAUNCH4(FR, 2) : GEMM3_LAUNCH2(FR, 2); } \
            return big ? GEMM3_LAUNCH4(FR, 1) : GEMM3_LAUNCH2(FR, 1);                                          \
        }                                                                                                      \
        return big ? GEMM3_LAUNCH4(FR, 0) : GEMM3_LAUNCH2(FR, 0);
        if constexpr (!A_KC || !B_KC) {
            if (arith == DLRM_ARITH_F32 && frag) {
                if (tiny) return launch_gemm3<A_KC, B_KC, 1, 0, 1, 1>(g, splits, st);
                if (small) return launch_gemm3<A_KC, B_KC, 1, 0, 1>(g, splits, st);
                GEMM3_BIG_OR_NOT(1)
            }
        }
        if (tiny) return launch_gemm3<A_KC, B_KC, 1, 0, 0, 1>(g, splits, st);
        if (small) return launch_gemm3<A_KC, B_KC, 1, 0>(g, splits, st);
        if (arith == DLRM_ARITH_BF16X6)
            return big ? launch_gemm3<A_KC, B_KC, 4, 1>(g, splits, st) : launch_gemm3<A_KC, B_KC, 2, 1>(g, splits, st);
        if (arith == DLRM_ARITH_BF16)
            return big ? launch_gemm3<A_KC, B_KC, 4, 2>(g, splits, st) : launch_gemm3<A_KC, B_KC, 2, 2>(g, splits, st);
        GEMM3_BIG_OR_NOT(0)
#undef GEMM3_BIG_OR_NOT
#undef GEMM3_LAUNCH2
#undef GEMM3_LAUNCH4
    }
    g.bits_in = nullptr; g.bits_out = nullptr;       // the any-shape kernel neither reads nor writes sign bits
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    static int dbg = -1;
    if (dbg < 0) dbg = DLRM_DEBUG_ENV("DLRM_GEMM_DEBUG", 0x7fffffff);
    g.debug = dbg;
    const size_t lds = 2 * 2 * TILE_F * sizeof(float);   // 73,728 B: two workgroups per CU
    static bool attr_done[DLRM_MAX_DEVICES] = {};
    const int dev = dlrm_current_device();
    if (!attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_f32_kernel<A_KC, B_KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[dev] = true;
    }
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits), block(256);
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC>), grid, block, lds, st, g);
    DLRM_LAUNCH_CHECK();
    return 0;
}

static bool arith_ok(int a) { return a == DLRM_ARITH_F32 || a == DLRM_ARITH_BF16X6 || a == DLRM_ARITH_BF16; }
static int vec_ok_kc(const float* p, long long ld, long long kext) { return dlrm_aligned16(p) && ld % 4 == 0 && kext % 4 == 0; }
// k-strided operand: a 16-byte load may run past the logical column extent as long as it stays inside the
// row pitch (ld % 4 == 0 guarantees that); the extra columns only feed output rows/columns >= M/N, which
// the epilogue never stores.
static int vec_ok_ks(const float* p, long long ld, long long cext) { (void)cext; return dlrm_aligned16(p) && ld % 4 == 0; }

}  // namespace

static int relu_bits_from(int64_t M, int N, const float* Y, int64_t ldy, uint64_t* bits, hipStream_t st) {
    const long long nblk = ((long long)N + 63) / 64, words = ((M + 31) / 32) * nblk * 64;
    long long nb = (words + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(relu_bits_kernel, dim3((unsigned)nb), dim3(256), 0, st, (long long)M, N, Y, (long long)ldy,
                       (unsigned*)bits, nblk, words);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_linear_fwd(int64_t M, int N, int K, const float* X, int64_t ldx, const float* W,
                               int64_t ldw, const float* bias, int act, float* Y, int64_t ldy,
                               uint64_t* relu_bits, int arith, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !X || !W || !Y) return DLRM_E_ARG;
    if (!arith_ok(arith)) return DLRM_E_MODE;
    if (relu_bits && act != DLRM_ACT_RELU) return DLRM_E_MODE;
    if (ldx < K || ldw < K || ldy < N) return DLRM_E_ARG;
    if (act < DLRM_ACT_NONE || act > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    if (N == 1 && gemm_path() != 2) {                // matrix-vector layer: HBM streaming, not MFMA (gemv.hip)
        const int rc = dlrm_gemv_fwd(M, K, X, ldx, W, bias, act, Y, ldy, (hipStream_t)stream);
        if (rc != DLRM_GEMV_NOT_HANDLED) return (rc == 0 && relu_bits) ? relu_bits_from(M, N, Y, ldy, relu_bits, (hipStream_t)stream) : rc;
    }
    GemmArgs g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
    g.vecA = vec_ok_kc(X, ldx, K); g.vecB = vec_ok_kc(W, ldw, K);
    g.vecC = dlrm_aligned16(Y) && ldy % 4 == 0;
    g.kchunk = ((K + BK - 1) / BK) * BK;
    g.bias = bias; g.act = act;
    g.bits_out = (unsigned*)relu_bits; g.bits_nblk = ((long long)N + 63) / 64;
    bool fast = false;
    const int rc = launch_gemm<true, true>(g, 1, (hipStream_t)stream, arith, &fast);
    if (rc == 0 && relu_bits && !fast) return relu_bits_from(M, N, Y, ldy, relu_bits, (hipStream_t)stream);
    return rc;
}

extern "C" int dlrm_cast_bf16(int64_t M, int N, int Npad, const float* src, int64_t lds_, uint16_t* dst, int64_t ldd, void* stream) {
    if (M <= 0 || N <= 0 || Npad < N || (Npad & 1) || !src || !dst || lds_ < N || ldd < Npad || (ldd & 1) || (((uintptr_t)dst) & 3u)) return DLRM_E_ARG;
    if (N % 8 == 0 && Npad % 8 == 0 && lds_ % 4 == 0 && ldd % 8 == 0 && dlrm_aligned16(src) && dlrm_aligned16(dst)) {
        long long nb = (M * (Npad / 8) + 255) / 256; if (nb > 0x7fffffffll) nb = 0x7fffffffll;
        hipLaunchKernelGGL(cast_bf16_vec8_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long long)M, N, Npad, src, (long long)lds_,
                           (unsigned short*)dst, (long long)ldd);
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    long long nb = (M * (Npad / 2) + 255) / 256; if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long long)M, N, Npad, src, (long long)lds_,
                       (unsigned short*)dst, (long long)ldd);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_cast_bf16_multi(int n, const float* const* src, const int64_t* lds_, const int* R, const int* C, uint16_t* const* dst,
                                    const int64_t* ldd, const int* Cpad, uint16_t* const* dstT, const int64_t* lddT, const int* Rpad, void* stream) {
    if (n <= 0 || n > DLRM_CAST_MULTI_MAX || !src || !lds_ || !R || !C || !dst || !ldd || !Cpad || !dstT || !lddT || !Rpad) return DLRM_E_ARG;
    CastMultiArgs a = {};
    a.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || R[i] <= 0 || C[i] <= 0 || lds_[i] < C[i] || (!dst[i] && !dstT[i])) return DLRM_E_ARG;
        if (dst[i] && (Cpad[i] < C[i] || ldd[i] < Cpad[i])) return DLRM_E_ARG;
        if (dstT[i] && (Rpad[i] < R[i] || lddT[i] < Rpad[i])) return DLRM_E_ARG;
        a.src[i] = src[i]; a.lds[i] = lds_[i]; a.R[i] = R[i]; a.C[i] = C[i];
        a.dst[i] = (unsigned short*)dst[i]; a.ldd[i] = ldd[i]; a.Cpad[i] = dst[i] ? Cpad[i] : 0;
        a.dstT[i] = (unsigned short*)dstT[i]; a.lddT[i] = lddT[i]; a.Rpad[i] = dstT[i] ? Rpad[i] : 0;
        const int rows = a.Rpad[i] > R[i] ? a.Rpad[i] : R[i], cols = a.Cpad[i] > C[i] ? a.Cpad[i] : C[i];
        a.tiles_c[i] = (cols + 31) / 32;
        a.tile_start[i] = total;
        total += ((rows + 31) / 32) * a.tiles_c[i];
    }
    for (int i = n; i <= DLRM_CAST_MULTI_MAX; ++i) a.tile_start[i] = total;
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_cast_bf16_transposed(int R, int C, int Rpad, const float* src, int64_t lds_, uint16_t* dstT, int64_t ldd, void* stream) {
    if (R <= 0 || C <= 0 || Rpad < R || !src || !dstT || lds_ < C || ldd < Rpad) return DLRM_E_ARG;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((Rpad + 31) / 32));
    hipLaunchKernelGGL(cast_bf16_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, R, C, Rpad, src, (long long)lds_, (unsigned short*)dstT,
                       (long long)ldd);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_split_bf16x3(int64_t M, int N, int Npad, const float* src, int64_t lds_, uint16_t* dst, int64_t ldd, int64_t plane_stride,
                                 void* stream) {
    if (M <= 0 || N <= 0 || Npad < N || !src || !dst || lds_ < N || ldd < Npad || plane_stride < (M - 1) * ldd + Npad) return DLRM_E_ARG;
    if (Npad % 8 || ldd % 8 || plane_stride % 8 || !dlrm_aligned16(dst)) return DLRM_E_ALIGN;
    long long nb = (M * (Npad / 8) + 255) / 256; if (nb > 0x7fffffffll) nb = 0x7fffffffll;
    if (N % 8 == 0 && lds_ % 4 == 0 && dlrm_aligned16(src))
        hipLaunchKernelGGL(split_bf16x3_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long long)M, N, Npad, src, (long long)lds_,
                           (unsigned short*)dst, (long long)ldd, (long long)plane_stride);
    else
        hipLaunchKernelGGL(split_bf16x3_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long long)M, N, Npad, src, (long long)lds_,
                           (unsigned short*)dst, (long long)ldd, (long long)plane_stride);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_split_bf16x3_transposed(int R, int C, int Rpad, const float* src, int64_t lds_, uint16_t* dstT, int64_t ldd, int64_t plane_stride,
                                            void* stream) {
    if (R <= 0 || C <= 0 || Rpad < R || !src || !dstT || lds_ < C || ldd < Rpad || plane_stride < (int64_t)(C - 1) * ldd + Rpad) return DLRM_E_ARG;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((Rpad + 31) / 32));
    hipLaunchKernelGGL(split_bf16x3_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, R, C, Rpad, src, (long long)lds_, (unsigned short*)dstT,
                       (long long)ldd, (long long)plane_stride);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// fp32-class product from PRE-SPLIT operands (arith "bf16x6"): C[M, N] (fp32, nullable) and / or Cp (three bf16 planes of the result, nullable)
// = epilogue(A . B^T), A [M, K] and B [N, K] given as three bf16 planes each (dlrm_split_bf16x3 / the Cp output of an earlier call), six
// v_mfma_f32_32x32x16_bf16 per 16 k in the order of the in-loop kernel: bit-identical to dlrm_linear_fwd / _bwd_data with DLRM_ARITH_BF16X6.
// dlrm_gemm_bf16x6_supported() == 0 (or DLRM_E_ALIGN here): the shape is outside the kernel's preconditions and the caller keeps fp32 storage.
extern "C" int dlrm_gemm_bf16x6_supported(int64_t M, int N, int K, int64_t lda, int64_t ldb) { return dlrm_gemm_bf16x6_ok(M, N, K, lda, ldb) ? 1 : 0; }

extern "C" int dlrm_gemm_bf16x6(int64_t M, int N, int K, const uint16_t* A, int64_t lda, int64_t planeA, const uint16_t* B, int64_t ldb, int64_t planeB,
                                const float* bias, int act, uint64_t* relu_bits_out, const uint64_t* relu_bits_in, float* C, int64_t ldc, uint16_t* Cp,
                                int64_t ldcp, int64_t planeC, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || (!C && !Cp)) return DLRM_E_ARG;
    if (lda < K || ldb < K || (C && ldc < N) || (Cp && (ldcp < N || planeC < (M - 1) * ldcp + N))) return DLRM_E_ARG;
    if (planeA < (M - 1) * lda + K || planeB < (int64_t)(N - 1) * ldb + K) return DLRM_E_ARG;
    if (act < DLRM_ACT_NONE || act > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    if (relu_bits_out && act != DLRM_ACT_RELU) return DLRM_E_MODE;
    if (!dlrm_aligned16(A) || !dlrm_aligned16(B) || (C && (!dlrm_aligned16(C) || ldc % 4)) || (Cp && (((uintptr_t)Cp) & 7u))) return DLRM_E_ALIGN;
    return dlrm_gemm_bf16x6_phased(M, N, K, A, lda, planeA, B, ldb, planeB, bias, act, relu_bits_out, relu_bits_in, C, ldc, Cp, ldcp, planeC,
                                   (hipStream_t)stream);
}

// C[M, N] (fp32, nullable) and / or Cb[M, N] (bf16, nullable) = epilogue(A[M, K] . B[N, K]^T), A and B bf16 in memory, fp32 accumulation:
//   forward        A = X, B = W:    + bias, activation, optional ReLU sign bits OUT (relu_bits_out)
//   data gradient  A = dY, B = W^T: result masked by the previous layer's ReLU sign bits (relu_bits_in)
// Preconditions (DLRM_E_ALIGN otherwise — the caller falls back to the fp32-storage kernels): K % 32 == 0, N % 4 == 0, 16-byte aligned
// operand rows (lda, ldb % 8 == 0), 16-byte aligned fp32 rows (ldc % 4 == 0), 8-byte aligned bf16 rows (ldcb % 4 == 0).
extern "C" int dlrm_gemm_bf16(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                              int act, uint64_t* relu_bits_out, const uint64_t* relu_bits_in, const float* addend, int64_t ldadd,
                              const float* addend2, int64_t ldadd2, float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || (!C && !Cb)) return DLRM_E_ARG;
    if (lda < K || ldb < K || (C && ldc < N) || (Cb && ldcb < N)) return DLRM_E_ARG;
    if (act < DLRM_ACT_NONE || act > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    if (relu_bits_out && act != DLRM_ACT_RELU) return DLRM_E_MODE;
    if (K % 32 || N % 4 || lda % 8 || ldb % 8 || !dlrm_aligned16(A) || !dlrm_aligned16(B) || (C && (!dlrm_aligned16(C) || ldc % 4)) ||
        (Cb && ((((uintptr_t)Cb) & 7u) || ldcb % 4)))
        return DLRM_E_ALIGN;
    {   // the bf16-shaped kernel (gemm_bf16.hip: 256 x 256 x 64 tile, four phases per k-tile) wherever its preconditions hold
        if (addend && (ldadd < N || ldadd % 4 || !dlrm_aligned16(addend))) return DLRM_E_ALIGN;
        if (addend2 && (!addend || ldadd2 < N || ldadd2 % 4 || !dlrm_aligned16(addend2))) return DLRM_E_ALIGN;
        const int rc = dlrm_gemm_bf16_phased(M, N, K, A, lda, B, ldb, bias, act, relu_bits_out, relu_bits_in, addend, ldadd, addend2, ldadd2, C, ldc,
                                             Cb, ldcb, (hipStream_t)stream);
        if (rc != DLRM_GEMV_NOT_HANDLED) return rc;
        if (addend) return DLRM_E_MODE;                 // the fp32-shaped kernel has no addend operand
    }
    GemmArgs g = {};
    g.M = M; g.N = N; g.K = K / 2;                    // in units of one fp32 word = two bf16 values
    g.A = (const float*)A; g.lda = lda / 2; g.B = (const float*)B; g.ldb = ldb / 2;
    g.C = C; g.ldc = ldc; g.Cb = (unsigned short*)Cb; g.ldcb = ldcb;
    g.vecA = 1; g.vecB = 1; g.vecC = 1;
    g.kchunk = g.K;
    g.bias = bias; g.act = act;
    g.bits_out = (unsigned*)relu_bits_out; g.bits_in = (const unsigned*)relu_bits_in; g.bits_nblk = ((long long)N + 63) / 64;
    const long long wg256 = ((M + 255) / 256) * ((N + 127) / 128);
    const bool big = M >= 256 && wg256 > 384 && wg256 <= 512;
    return big ? launch_gemm3<true, true, 4, 3>(g, 1, (hipStream_t)stream) : launch_gemm3<true, true, 2, 3>(g, 1, (hipStream_t)stream);
}

// DCN-v2 cross layer, second product WITH its Hadamard half (torchrec LowRankCrossNet, torchrec_dlrm/dlrm_main.py:608-619):
//   u = A . B^T + bias (A = v_l bf16 [M, K], B = W_l bf16 [N, K]);  Ub (bf16, nullable) = u;  C (fp32) = fma(x0, u, xl);  Cb (bf16, nullable) = bf16(C)
// — what dlrm_gemm_bf16 + dlrm_cross_fwd computed in two passes (the fp32 u written by one and re-read by the other), same operation order.
// Only where the bf16-shaped kernel runs (DLRM_E_MODE otherwise: the caller keeps the two kernels).
extern "C" int dlrm_gemm_bf16_cross(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                                    const float* x0, int64_t ldx0, const float* xl, int64_t ldxl, uint16_t* Ub, int64_t ldub,
                                    float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !x0 || !xl || !C) return DLRM_E_ARG;
    if (lda < K || ldb < K || ldc < N || ldx0 < N || ldxl < N || (Cb && ldcb < N) || (Ub && ldub < N)) return DLRM_E_ARG;
    if (K % 32 || N % 4 || lda % 8 || ldb % 8 || !dlrm_aligned16(A) || !dlrm_aligned16(B) || !dlrm_aligned16(C) || ldc % 4 ||
        !dlrm_aligned16(x0) || ldx0 % 4 || !dlrm_aligned16(xl) || ldxl % 4 || (Cb && ((((uintptr_t)Cb) & 7u) || ldcb % 4)) ||
        (Ub && ((((uintptr_t)Ub) & 7u) || ldub % 4)))
        return DLRM_E_ALIGN;
    const int rc = dlrm_gemm_bf16_phased(M, N, K, A, lda, B, ldb, bias, DLRM_ACT_NONE, nullptr, nullptr, xl, ldxl, nullptr, 0, C, ldc, Cb, ldcb,
                                         (hipStream_t)stream, x0, ldx0, Ub, ldub);
    return rc == DLRM_GEMV_NOT_HANDLED ? DLRM_E_MODE : rc;
}

extern "C" int64_t dlrm_relu_bits_bytes(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return ((M + 31) / 32) * (((int64_t)N + 63) / 64) * 32 * 8;
}

extern "C" int dlrm_linear_bwd_data(int64_t M, int N, int K, const float* dY, int64_t lddy,
                                    const float* W, int64_t ldw, const float* Xact, int64_t ldxa,
                                    int xact_kind, const uint64_t* relu_bits, float* dX, int64_t lddx, int arith, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !dY || !W || !dX) return DLRM_E_ARG;
    if (!arith_ok(arith)) return DLRM_E_MODE;
    if (lddy < N || ldw < K || lddx < K) return DLRM_E_ARG;
    if (xact_kind < DLRM_ACT_NONE || xact_kind > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    if (xact_kind != DLRM_ACT_NONE && (!Xact || ldxa < K)) return DLRM_E_ARG;
    if (N == 1 && gemm_path() != 2) {
        const int rc = dlrm_gemv_bwd_data(M, K, dY, lddy, W, xact_kind != DLRM_ACT_NONE ? Xact : nullptr, ldxa, xact_kind, dX,
                                          lddx, (hipStream_t)stream);
        if (rc != DLRM_GEMV_NOT_HANDLED) return rc;
    }
    GemmArgs g = {};
    g.M = M; g.N = K; g.K = N;                       // output [M, K_layer], reduce over N_layer
    g.A = dY; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = lddx;
    g.vecA = vec_ok_kc(dY, lddy, N); g.vecB = vec_ok_ks(W, ldw, K);
    g.vecC = dlrm_aligned16(dX) && lddx % 4 == 0;
    g.kchunk = ((N + BK - 1) / BK) * BK;
    g.act = DLRM_ACT_NONE;
    if (xact_kind != DLRM_ACT_NONE) {
        g.mask = Xact; g.ldmask = ldxa; g.mask_act = xact_kind;
        if (relu_bits && xact_kind == DLRM_ACT_RELU) {       // sign bits written by the forward pass: no fp32 mask read (fast path only)
            g.bits_in = (const unsigned*)relu_bits; g.bits_nblk = ((long long)K + 63) / 64;
        } else {
            g.vecC = g.vecC && dlrm_aligned16(Xact) && ldxa % 4 == 0;
        }
    }
    return launch_gemm<true, false>(g, 1, (hipStream_t)stream, arith);
}

static void wgrad_plan(int64_t M, int N, int K, int* splits_out, int64_t* kchunk_out) {
    // split the batch reduction so that ~4 workgroups per CU exist (1024: with the two-stage ring four 128-row workgroups fit a CU, and the wide
    // layers then take the 256-row tiles at exactly one round of 512 — round 6, in-step A/B profiles/round6/gemm_nst_wgrad_plan_ab.md: 768 -> 1024 =
    // linear_bwd_weight 2.44 -> 2.37 ms per step; round 5 had gone 1024 -> 768 with the three-stage ring); slices of >= 512 rows.
    // NOTE: the plan fixes the split-K summation order: weight gradients differ in the last bits between plans (deterministic for one plan)
    const int tiles = (int)(((N + BM - 1) / BM) * ((K + BN - 1) / BN));
    // tuning aids: DLRM_WGRAD_WGS (workgroups a launch aims at), DLRM_WGRAD_MINROWS (shortest batch slice)
    static const int target_wgs_e = DLRM_TUNE_ENV("DLRM_WGRAD_WGS", DLRM_WGRAD_WGS_DEFAULT), min_rows_e = DLRM_TUNE_ENV("DLRM_WGRAD_MINROWS", 512);
    const int target_wgs = target_wgs_e > 0 ? target_wgs_e : DLRM_WGRAD_WGS_DEFAULT, min_rows = min_rows_e >= 128 ? min_rows_e : 512;
    int splits = (target_wgs + tiles - 1) / tiles;
    // few output tiles (512 -> 256: 8): every slice costs a whole slab of N x K floats written and read back, so no more slices than
    // keep >= 2 workgroups per CU (round 6: 64 x 8 instead of 96 x 8 = 146 -> 139 us per call; tuning builds: DLRM_WGRAD_MAXSPLITS, 0 = no cap)
    static const int cap_e = DLRM_TUNE_ENV("DLRM_WGRAD_MAXSPLITS", DLRM_WGRAD_MAXSPLITS_DEFAULT);
    if (cap_e > 0 && splits > cap_e && (long long)tiles * cap_e >= 512) splits = cap_e;
    const int64_t max_splits = (M + min_rows - 1) / min_rows;
    if (splits > max_splits) {
        // small batches (Criteo-Kaggle: 2048 rows): slices down to 128 rows while the launch still has fewer workgroups than the chip has CUs
        const int64_t max128 = (M + 127) / 128;
        int64_t want = (256 + tiles - 1) / tiles;
        if (want > max128) want = max128;
        splits = (int)(want > max_splits ? want : max_splits);
    }
    if (splits < 1) splits = 1;
    int64_t kchunk = (M + splits - 1) / splits;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    *splits_out = (int)((M + kchunk - 1) / kchunk);
    *kchunk_out = kchunk;
}

extern "C" int64_t dlrm_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int splits; int64_t kchunk;
    wgrad_plan(M, N, K, &splits, &kchunk);
    const int64_t ldp = ((int64_t)K + 3) & ~(int64_t)3;
    int64_t need = (int64_t)(splits < 1 ? 1 : splits) * N * (ldp + 1) * (int64_t)sizeof(float);   // >= one slab (the padded-dW form needs it even unsplit) + the k-slices' bias-gradient row sums
    if (N == 1) {                                    // the matrix-vector path keeps per-workgroup column partials
        const int64_t gv = dlrm_gemv_bwd_weight_workspace_bytes(M, K);
        if (gv > need) need = gv;
    }
    if (K <= 16 && M >= 4096) {
        const int64_t sk = dlrm_smallk_bwd_weight_workspace_bytes(M, N, K);
        if (sk > need) need = sk;
    }
    return need;
}

// K_store <= K: dW is [N, K_store]; the columns K_store..K-1 of X are alignment padding (zeros) whose gradient is dropped
static int linear_bwd_weight_impl(int64_t M, int N, int K, int K_store, const float* dY, int64_t lddy,
                                  const float* X, int64_t ldx, float* dW, int64_t lddw,
                                  float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                  int arith, void* stream) {
    if (!arith_ok(arith)) return DLRM_E_MODE;
    if (M <= 0 || N <= 0 || K <= 0 || K_store <= 0 || K_store > K || !dY || !X || !dW) return DLRM_E_ARG;
    if (lddy < N || ldx < K || lddw < K_store) return DLRM_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (N == 1 && K_store == K && gemm_path() != 2) {
        const int rc = dlrm_gemv_bwd_weight(M, K, dY, lddy, X, ldx, dW, dbias, accumulate, workspace, workspace_bytes, st);
        if (rc != DLRM_GEMV_NOT_HANDLED) return rc;
    }
    if (K <= 16 && M >= 4096 && gemm_path() != 2) {
        const int rc = dlrm_smallk_bwd_weight(M, N, K, K_store, dY, lddy, X, ldx, dW, lddw, dbias, accumulate, workspace, workspace_bytes, st);
        if (rc != DLRM_GEMV_NOT_HANDLED) return rc;
    }
    GemmArgs g = {};
    g.M = N; g.N = K; g.K = M;                       // output [N_layer, K_layer], reduce over the batch
    g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx; g.C = dW; g.ldc = lddw;
    g.vecA = vec_ok_ks(dY, lddy, N); g.vecB = vec_ok_ks(X, ldx, K);
    g.vecC = dlrm_aligned16(dW) && lddw % 4 == 0;
    g.act = DLRM_ACT_NONE;
    g.rowsumA = dbias;                               // db[n] = sum_m dY[m, n], from the A fragments
    int splits; int64_t kchunk;
    wgrad_plan(M, N, K, &splits, &kchunk);
    g.kchunk = kchunk;
    // split-K partials: with a workspace every k-slice stores its own [N, ldp] slab with plain 16-byte
    // stores and a second kernel sums the slabs in a fixed order (deterministic; ~2 x splits x N x K x 4 bytes
    // of streaming traffic).  Without one the slices accumulate into dW with fp32 atomics (measured 2-3x
    // slower on the small-output layers: up to 128 atomic adds land on every address).
    const int64_t ldp = ((int64_t)K + 3) & ~(int64_t)3;
    const int64_t slab = (int64_t)N * ldp;
    const bool use_ws = (splits > 1 || K_store < K) && workspace && dlrm_aligned16(workspace) &&
                        workspace_bytes >= (int64_t)splits * (slab + N) * (int64_t)sizeof(float);
    if (K_store < K && !use_ws) return DLRM_E_ARG;   // a narrower dW needs the slab path (pass the queried workspace)
    if (use_ws) {
        g.C = (float*)workspace; g.ldc = ldp; g.vecC = 1; g.c_split_stride = slab; g.atomic_out = 0;
        float* rs_part = dbias ? (float*)workspace + (int64_t)splits * slab : nullptr;
        if (dbias) { g.rowsumA = rs_part; g.rowsum_split_stride = N; }
        int rc = launch_gemm<false, false>(g, splits, st, arith);
        if (rc) return rc;
        const bool v4 = dlrm_aligned16(dW) && lddw % 4 == 0 && K_store % 4 == 0;
        const long long items = (long long)N * (v4 ? K_store / 4 : K_store);
        int blocks = (int)((items + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
        if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, N, K_store, splits, (const float*)workspace,
                                   (long long)ldp, (long long)slab, dW, (long long)lddw, accumulate ? 1 : 0, (const float*)rs_part, dbias);
        else    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, N, K_store, splits, (const float*)workspace,
                                   (long long)ldp, (long long)slab, dW, (long long)lddw, accumulate ? 1 : 0, (const float*)rs_part, dbias);
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    g.atomic_out = (splits > 1 || accumulate) ? 1 : 0;
    if (!accumulate) {
        int e = 0;                            // zeroed by kernels (no memset nodes in a captured step, common.h)
        if (g.atomic_out) e = dlrm_zero2d(dW, lddw, K, N, st);
        if (e == 0 && dbias) e = dlrm_zero2d(dbias, N, N, 1, st);
        if (e) return e;
    }
    return launch_gemm<false, false>(g, splits, st, arith);
}

// The backward of an N == 1 layer (the 256 -> 1 head of the top tower, dlrm_s_pytorch.py:208-246 / AddmmBackward + SigmoidBackward :1613) in one
// pass: dz = dY * act'(Y), dW = dz^T X, db = sum dz, dX = (dz W) * xact'(X).  DLRM_E_MODE: shapes outside csrc/gemv.hip's fast path (the caller
// runs dlrm_act_bwd + dlrm_linear_bwd_weight + dlrm_linear_bwd_data, whose bits these are).
extern "C" int dlrm_linear_head_bwd(int64_t M, int K, const float* dY, int64_t lddy, const float* Y, int64_t ldy, int act,
                                    const float* X, int64_t ldx, const float* W, int xact_kind, float* dX, int64_t lddx,
                                    float* dW, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M <= 0 || K <= 0 || !dY || !X || !W || !dW) return DLRM_E_ARG;
    if (lddy < 1 || ldx < K || (dX && lddx < K) || (Y && ldy < 1)) return DLRM_E_ARG;
    if (act < DLRM_ACT_NONE || act > DLRM_ACT_SIGMOID || xact_kind < DLRM_ACT_NONE || xact_kind > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    const int rc = dlrm_gemv_bwd_fused(M, K, dY, lddy, (Y && act != DLRM_ACT_NONE) ? Y : nullptr, ldy, act, X, ldx, W, xact_kind, dX, lddx, dW,
                                       dbias, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
    return rc == DLRM_GEMV_NOT_HANDLED ? DLRM_E_MODE : rc;
}

extern "C" int dlrm_linear_bwd_weight(int64_t M, int N, int K, const float* dY, int64_t lddy,
                                      const float* X, int64_t ldx, float* dW, int64_t lddw,
                                      float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                      int arith, void* stream) {
    return linear_bwd_weight_impl(M, N, K, K, dY, lddy, X, ldx, dW, lddw, dbias, accumulate, workspace, workspace_bytes, arith, stream);
}

extern "C" int dlrm_linear_bwd_weight_padded(int64_t M, int N, int K, int K_store, const float* dY, int64_t lddy,
                                             const float* X, int64_t ldx, float* dW, int64_t lddw,
                                             float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                             int arith, void* stream) {
    return linear_bwd_weight_impl(M, N, K, K_store, dY, lddy, X, ldx, dW, lddw, dbias, accumulate, workspace, workspace_bytes, arith, stream);
}

// Weight gradient of the bf16 tower from bf16 operands AS STORED: dW[N, K] (+)= dZ[M, N]^T . X[M, K], db[N] (+)= column sums of dZ — the
// reference's AddmmBackward / sum over the batch (dlrm_s_pytorch.py:1613) in the arithmetic of dlrm_gemm_bf16 (bf16 products, fp32 accumulation).
// Both operands are k-STRIDED (the batch is the reduction): csrc/gemm_bf16.hip reads them through ds_read_b64_tr_b16, split over the batch
// into fp32 slabs that splitk_reduce_kernel sums in slice order (deterministic dW and db, no atomics, no zero fill).
// DLRM_E_ALIGN when the shape is outside the kernel's preconditions (M % 64, N % 8, K % 8, widths >= 64, 16-byte aligned rows): the caller
// keeps dlrm_linear_bwd_weight on fp32 operands.
extern "C" int64_t dlrm_linear_bwd_weight_bf16_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int splits; int64_t kchunk;
    dlrm_gemm_bf16_wgrad_plan(M, N, K, &splits, &kchunk);
    const int64_t ldp = ((int64_t)K + 3) & ~(int64_t)3;
    return (int64_t)splits * N * (ldp + 1) * (int64_t)sizeof(float);
}

static int linear_bwd_weight_planes(int64_t M, int N, int K, int K_store, const uint16_t* dZ, int64_t lddz, const uint16_t* X, int64_t ldx,
                                    float* dW, int64_t lddw, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                    void* stream, int planes, int64_t planeZ, int64_t planeX) {
    if (M <= 0 || N <= 0 || K <= 0 || K_store <= 0 || K_store > K || !dZ || !X || !dW || !workspace) return DLRM_E_ARG;
    if (lddz < N || ldx < K || lddw < K_store) return DLRM_E_ARG;
    if (!dlrm_gemm_bf16_wgrad_ok(M, N, K, lddz, ldx) || !dlrm_aligned16(dZ) || !dlrm_aligned16(X) || !dlrm_aligned16(workspace)) return DLRM_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    int splits; int64_t kchunk;
    dlrm_gemm_bf16_wgrad_plan(M, N, K, &splits, &kchunk);
    const int64_t ldp = ((int64_t)K + 3) & ~(int64_t)3, slab = (int64_t)N * ldp;
    if (workspace_bytes < (int64_t)splits * (slab + N) * (int64_t)sizeof(float)) return DLRM_E_ARG;
    float* rs_part = dbias ? (float*)workspace + (int64_t)splits * slab : nullptr;
    const int rc = dlrm_gemm_bf16_wgrad_phased(M, N, K, dZ, lddz, X, ldx, (float*)workspace, ldp, slab, rs_part, splits, kchunk, st, planes, planeZ, planeX);
    if (rc) return rc;
    // dW may be NARROWER than the product (K_store < K: the trailing columns of X are zero padding whose gradient is dropped)
    const bool v4 = dlrm_aligned16(dW) && lddw % 4 == 0 && K_store % 4 == 0;
    const long long items = (long long)N * (v4 ? K_store / 4 : K_store);
    int blocks = (int)((items + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, N, K_store, splits, (const float*)workspace,
                               (long long)ldp, (long long)slab, dW, (long long)lddw, accumulate ? 1 : 0, (const float*)rs_part, dbias);
    else    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, N, K_store, splits, (const float*)workspace,
                               (long long)ldp, (long long)slab, dW, (long long)lddw, accumulate ? 1 : 0, (const float*)rs_part, dbias);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_linear_bwd_weight_bf16(int64_t M, int N, int K, int K_store, const uint16_t* dZ, int64_t lddz, const uint16_t* X, int64_t ldx,
                                           float* dW, int64_t lddw, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                                           void* stream) {
    return linear_bwd_weight_planes(M, N, K, K_store, dZ, lddz, X, ldx, dW, lddw, dbias, accumulate, workspace, workspace_bytes, stream, 0, 0, 0);
}

// the same from three bf16 planes per operand (arith "bf16x6"; workspace: dlrm_linear_bwd_weight_bf16_workspace_bytes)
extern "C" int dlrm_linear_bwd_weight_bf16x6(int64_t M, int N, int K, int K_store, const uint16_t* dZ, int64_t lddz, int64_t planeZ, const uint16_t* X,
                                             int64_t ldx, int64_t planeX, float* dW, int64_t lddw, float* dbias, int accumulate, void* workspace,
                                             int64_t workspace_bytes, void* stream) {
    if (planeZ < (M - 1) * lddz + N || planeX < (M - 1) * ldx + K) return DLRM_E_ARG;
    return linear_bwd_weight_planes(M, N, K, K_store, dZ, lddz, X, ldx, dW, lddw, dbias, accumulate, workspace, workspace_bytes, stream, 1, planeZ, planeX);
}

extern "C" int dlrm_act_bwd(int64_t M, int N, const float* dY, int64_t lddy, const float* Y,
                            int64_t ldy, int act, float* dZ, int64_t lddz, float* dbias, void* stream) {
    if (M <= 0 || N <= 0 || !dY || !Y || !dZ) return DLRM_E_ARG;
    if (lddy < N || ldy < N || lddz < N) return DLRM_E_ARG;
    if (act < DLRM_ACT_NONE || act > DLRM_ACT_SIGMOID) return DLRM_E_MODE;
    int tx = pow2ceil_i(N); if (tx > 64) tx = 64;
    const int ty_n = 256 / tx;
    int rows_per_block = ty_n * 16;
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), (unsigned)((N + tx - 1) / tx), 1);
    hipLaunchKernelGGL(act_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (long long)M, N, dY,
                       (long long)lddy, Y, (long long)ldy, act, dZ, (long long)lddz, dbias, tx, rows_per_block);
    DLRM_LAUNCH_CHECK();
    return 0;
}
