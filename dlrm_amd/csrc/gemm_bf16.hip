// gemm_bf16.hip — the bf16-SHAPED GEMM of the bf16 MLP towers (BASELINE.json configs[4], "bf16 MLP on MFMA";
// torchrec_dlrm/README.MD:177-194, dlrm_main.py:526,598-619) for gfx950.
//
//   C[M, N] (fp32, optional) / Cb[M, N] (bf16, optional) = epilogue( A[M, K] . B[N, K]^T ),   A and B bf16, k contiguous
//     forward        A = X,  B = W      + bias, activation, ReLU sign bits OUT
//     data gradient  A = dY, B = W^T    masked by the previous layer's ReLU sign bits IN
//
// Why a second kernel beside gemm3_kernel<.., ARITH = 3> (gemm.hip): that one reuses the fp32-shaped pipeline — a 256 x 128 tile, one
// barrier and six LDS-DMA issues per 32 k for 16 MFMAs — and sits at 0.08-0.15 of the bf16 matrix peak (profiles/round3).  At bf16 rates
// a 256 x 128 x 32 step is 512 MFMA cycles per wave: the barrier, the DMA issue and the fragment reads all have to hide under it, which a
// load-all-then-multiply body cannot do.  This kernel is shaped for the bf16 pipe:
//   * tile 256 x 256 x 64, 8 waves (2 x 4) of 128 x 64 = 4 x 2 sub-tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers),
//     one workgroup per CU (two waves per SIMD), 128 KiB of LDS = 2 k-tiles x (A 32 KiB + B 32 KiB): 128 FLOP per staged byte;
//   * FOUR PHASES per k-tile, each = one quadrant (64 x 32 x 64: 8 MFMAs = 256 matrix-pipe cycles) of the wave tile:
//         LOAD  : ds_read_b128 of the quadrant's NEW fragments (8 A + 4 B, 4 B, 8 A, 4 B), 2 LDS-DMA issues of the NEXT k-tile, counted vmcnt
//         s_barrier
//         MATH  : 8 MFMAs at raised priority
//         s_barrier
//     the two wave halves (rows 0-127 / 128-255 of the tile; one wave of each half per SIMD) run ONE BARRIER APART, so on every SIMD
//     one wave multiplies while the other reads fragments and issues DMA — fragment reads of phase p+1 under the MFMAs of phase p;
//   * the next k-tile is staged in four 16 KiB pieces in the ORDER OF FIRST USE (A rows of the first two sub-tile rows, B columns of the
//     first sub-tile column, B second column, A last two rows), one piece per phase: every piece is issued >= 3 phases before its first
//     read and its slot is rewritten >= 2 phases after its last read; `s_waitcnt vmcnt(4)` leaves two pieces in flight across every barrier;
//   * LDS image per operand and k-half: [256 rows][64 B] with the 16-byte slots XOR-swizzled through the SOURCE address (the DMA
//     destination is lane-linear) — the image gemm3_kernel uses, conflict-free for the 32-row x 16-byte fragment reads.  (A [256 rows][128 B]
//     image fed by full-line DMA chunks of 8 rows x 128 B measured 4-10 % SLOWER, profiles/round4/bf16_gemm_notes.md);
//   * epilogue as in gemm3_kernel: accumulators hold the TRANSPOSED sub-tiles, bands pass through wave-private LDS and leave as 16-byte
//     (fp32) / 8-byte (bf16) row segments; bias, activation, sign bits out (one 4-byte word per lane and 32 x 64 block), sign-bit mask in.
// Preconditions (checked by the host; otherwise the caller keeps gemm3_kernel): K % 64 == 0, N % 4 == 0, N >= 192, M >= 256, 16-byte aligned
// operand rows.  Rows / columns past the matrix edge are clamped to a valid address and never stored.
//
// The same kernel template carries two more forms:
//   WG = true   the WEIGHT GRADIENT dW = dZ^T X from the stored bf16 operands (both k-strided: fragments by ds_read_b64_tr_b16, split-K slabs);
//   PL = 3      arith "bf16x6": every operand is THREE bf16 planes of an fp32 tensor, six MFMAs per 16 k (dlrm_gemm_bf16x6,
//               dlrm_linear_bwd_weight_bf16x6; K % 16 == 0) — see the comment at the kernel.
// Measured context for every rate quoted against "the bf16 peak": on random operands the matrix pipe of this part holds 1.88 of its nominal
// 2.46 PFLOP/s (bench.py box.mfma_bf16_random_tflops, profiles/round4/box_classes.md).
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

struct BfArgs {
    long long M, N, K;                       // K in bf16 elements, multiple of 64
    const unsigned short* A; long long lda;  // elements
    const unsigned short* B; long long ldb;
    float* C; long long ldc;                 // nullable
    unsigned short* Cb; long long ldcb;      // nullable
    const float* bias; int act;
    const float* addend; long long ldadd;    // nullable: fp32 [M, N] added to the result (after bias / activation / mask); may alias C
    const float* addend2; long long ldadd2;  // nullable: a second one (the DCN-v2 backward's last sum: dx0 + g + dv.V in one epilogue)
    // DCN-v2 cross layer with its Hadamard half in the epilogue (dlrm_gemm_bf16_cross): u = A.B^T + bias is stored as bf16 in Ub (nullable) and
    // the result becomes fma(mul, u, addend) = x0 * u + xl (the operation order of cross_fwd_kernel: bit-identical)
    const float* mul; long long ldmul;       // nullable; needs addend
    unsigned short* Ub; long long ldub;      // nullable
    unsigned* bits_out; const unsigned* bits_in; long long bits_nblk;
    int tiles_m, tiles_n;
    int wide16;                              // bf16-only output through 16-byte stores (set by the host when its preconditions hold)
    int debug;                               // tuning build only (make TUNING=1, env DLRM_BF16_DEBUG): 1 no DMA in the k-loop, 2 fragments read once, 4 no MFMAs, 8 no epilogue, 16 epilogue without its global stores — WRONG results
    // weight-gradient form (WG): the reduction runs over the ROWS of both operands (A = dZ [K, M], B = X [K, N], C = A^T B), split in
    // gridDim.z slices of kchunk rows; slice z stores its fp32 partial at C + z * c_split_stride and the row sums of A^T (the bias
    // gradient) at rowsum + z * M
    long long kchunk, c_split_stride;
    float* rowsum;
    // PL = 3 ("bf16x6": fp32 operands stored as three bf16 planes h, m, l with x == h + m + l exactly — split3 of gemm.hip): the planes of an
    // operand are [rows, ld] matrices planeA / planeB BYTES apart (one allocation: the DMA adds the plane to its 32-bit lane offset); the bf16
    // output Cb becomes three planes planeC ELEMENTS apart
    long long planeA, planeB, planeC;
};

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int OP_BYTES = 2 * 256 * 64;            // one operand of one k-tile: [2 k-halves][256 rows][64 B] = 32 KiB
constexpr int KHALF_BYTES = 256 * 64;             // 16 KiB
constexpr int STAGE_BYTES = 2 * OP_BYTES;         // 64 KiB
constexpr int P_EPI_LD = 64 + 4;

__device__ __forceinline__ void p_glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void p_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void p_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned p_cvt_pk_bf16(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// two fp32 values -> packed pairs of their three bf16 planes (truncation split, x == h + m + l exactly: split3 of gemm.hip)
__device__ __forceinline__ void p_split2(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ph = __builtin_amdgcn_perm(ub, ua, 0x07060302);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned ura = __float_as_uint(ra), urb = __float_as_uint(rb);
    pm = __builtin_amdgcn_perm(urb, ura, 0x07060302);
    const float sa = ra - __uint_as_float(ura & 0xffff0000u), sb = rb - __uint_as_float(urb & 0xffff0000u);
    pl = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);
}
__device__ __forceinline__ float p_act(float v, int act) {
    if (act == DLRM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DLRM_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
#define P_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)

template <int V> struct IC { static constexpr int value = V; };
typedef short shortx4 __attribute__((ext_vector_type(4)));
// two ds_read_b64_tr_b16 = the 8 consecutive k-values of one column a lane feeds to v_mfma_f32_32x32x16_bf16 from a [k][column] image:
// inside each 16-lane group lane p passes the address of [row p >> 2][columns 4 (p & 3) .. + 3] and lane i receives column i of rows 0..3
// (measured: tools/probes/tr_read_probe.hip, profiles/round4/tr_read_probe.txt)
__device__ __forceinline__ uintx4 p_tr_read8(unsigned lds_addr) {
    typedef __attribute__((address_space(3))) shortx4* lp;
    struct Pair { shortx4 lo, hi; } v;
    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)lds_addr);
    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_addr + 256));      // k-rows + 4
    return __builtin_bit_cast(uintx4, v);
}

// WG = false: C = A . B^T, both operands k-contiguous (forward, data gradient).
// WG = true : C[M, N] = sum over rows r of A[r, m] * B[r, n] — the weight gradient dW = dZ^T X with A = dZ [batch, out], B = X [batch, in],
//             both operands k-STRIDED.  LDS image of an operand and k-tile: 8 sub-runs of 32 columns, each [64 k-rows][64 B]; a 1 KiB DMA
//             chunk = 16 k-rows x 64 B of one sub-run (the same lane -> (row, 16-byte slot) map as the k-contiguous chunks, no swizzle:
//             a 32-lane ds_read_b64_tr_b16 touches 4 k-rows x 64 B = one whole 256-byte bank row); fragments by p_tr_read8.
//
// PL = 3: the SAME pipeline for fp32 operands held as three bf16 planes (arith "bf16x6", split once by the producer instead of in every
// k-loop that reads them: VERDICT r3 item 7).  A k-tile is 16 k of all three planes (A 24 KiB + B 24 KiB per stage): per quadrant the
// LOAD segment reads 2 x 3 A and 3 B fragments and the MATH segment issues the SIX products of gemm3_kernel<ARITH = 1> for each of its two
// accumulators, in that kernel's order (l.h, h.l, m.m, m.h, h.m, h.h) — every accumulator sees the same sequence of MFMAs as in the in-loop
// kernel, so the k-contiguous forms are BIT-IDENTICAL to it.  12 MFMAs per 9 fragment reads (PL = 1: 8 per 12): the LDS pipe, which
// co-limits the bf16 form, has 2.7x the slack here.  LDS image, k-contiguous: [plane][256 rows][32 B], the two 16-byte slots of a row
// swapped for rows 16-31 of every 32 (ds_read_b128 serves lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} together: each group then
// covers the 256-byte bank row once); k-strided (WG): [sub-run][plane][16 k-rows][64 B].  A piece = 12 chunks of 1 KiB (4 row groups x 3
// planes): every wave issues one, waves 0-3 a second one — the counted waits differ by wave accordingly.
// EP: 0 = the general epilogue; 1 / 2 = STRAIGHT-LINE bf16-only output (the lean towers' 16-byte stores), no / ReLU activation, no addends;
// 3 = straight-line fp32 store (the weight gradient's slabs).  Round 6, as in gemm3_kernel: the general body's run-time branches and in-loop
// addend loads make the compiler close every join with s_waitcnt vmcnt(0) — 108 of them for 116 stores in the forward instance — so the
// stores of a band left one memory round trip apart; the straight-line bodies contain no load and no divergent path.
template <bool WG, int PL, int EP = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_phased_kernel(BfArgs g) {
    static_assert(EP == 0 || (EP == 3) == WG, "EP 1 / 2: forward / data gradient; EP 3: weight gradient");
    constexpr int BK_ = PL == 1 ? PBK : 16;                       // k per tile
    constexpr int OPB = PL == 1 ? OP_BYTES : 3 * 256 * 32;        // one operand of one k-tile (PL 3: 24 KiB)
    constexpr int STG = 2 * OPB;
    constexpr int NJ = PL == 1 ? 4 : 3;                           // fragments per sub-tile and k-tile: four 16-k steps / three planes
    constexpr int SUBB = PL == 1 ? 4096 : 3072;                   // WG: bytes of one 32-column sub-run
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const char* ldsb = (const char*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;              // wave row (its half of the tile's rows), wave column
    const int l31 = lane & 31, h = lane >> 5;

    // workgroups go to the 8 XCDs round-robin in dispatch order: every XCD gets a contiguous range of tiles (the tiles_n tiles of one A
    // row panel meet in one L2)
    int id, zs = 0;
    {
        // (WG: the output tiles of one k-slice sit on ONE XCD and march through the same rows of dZ and X together, as in gemm3_kernel)
        const int nwg = g.tiles_m * g.tiles_n, total = nwg * (int)gridDim.z, lin = (int)blockIdx.z * nwg + (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, local = lin >> 3;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        zs = w / nwg; id = w - zs * nwg;
    }
    zs = __builtin_amdgcn_readfirstlane(zs);
    const int tile_m = __builtin_amdgcn_readfirstlane(id / g.tiles_n), tile_n = __builtin_amdgcn_readfirstlane(id - tile_m * g.tiles_n);
    const long long m0 = (long long)tile_m * PBM, n0 = (long long)tile_n * PBN;
    const long long k_begin = WG ? (long long)zs * g.kchunk : 0;
    const long long k_end = WG ? ((k_begin + g.kchunk < g.K) ? k_begin + g.kchunk : g.K) : g.K;
    const int nk = (int)((k_end - k_begin) / BK_);
    const bool second = PL == 1 || wave < 4;                      // this wave issues a second chunk of every piece

    // ---- DMA plan.  A piece = 128 tile rows (or columns) x 64 k = 16 chunks of 1 KiB; wave w moves chunks w and w + 8 of every piece.
    //   piece 0 (A0): A rows {0..63, 128..191}      piece 1 (B0): B rows {64 c + 0..31,  c = 0..3}
    //   piece 3 (A1): A rows {64..127, 192..255}    piece 2 (B1): B rows {64 c + 32..63, c = 0..3}
    unsigned voff[4][2], dst[4][2];
    if constexpr (WG) {
        // piece = 4 sub-runs x 4 chunks (16 k-rows each); wave w moves chunks w and w + 8: sub-run (c >> 2) of the piece, k-rows 16 (c & 3) ..
        //   A sub-run = wr * 4 + tm (32 columns of dZ each): A0 = {0,1,4,5}, A1 = {2,3,6,7};  B sub-run = wc * 2 + tn: B0 = {0,2,4,6}, B1 = {1,3,5,7}
        const int srow = lane >> 2, sslot = lane & 3;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = wave + 8 * i;
                const int sp = PL == 1 ? c >> 2 : c & 3, mc = PL == 1 ? c & 3 : 0, plane = PL == 1 ? 0 : c >> 2;      // (PL 3: chunk = (sub-run, plane), all 16 k-rows)
                int sub;
                if (p == 0)      sub = (sp >> 1) * 4 + (sp & 1);
                else if (p == 3) sub = (sp >> 1) * 4 + 2 + (sp & 1);
                else if (p == 1) sub = 2 * sp;
                else             sub = 2 * sp + 1;
                const bool isA = (p == 0 || p == 3);
                const long long c0 = isA ? m0 : n0, cmax = isA ? g.M : g.N, ld = isA ? g.lda : g.ldb;
                long long col = c0 + sub * 32 + sslot * 8; if (col > cmax - 8) col = cmax - 8;      // (extents are multiples of 8: clamped columns are never stored)
                voff[p][i] = (unsigned)((((long long)(mc * 16 + srow)) * ld + (col - c0)) * 2 + plane * (isA ? g.planeA : g.planeB));
                dst[p][i] = lds_base + (isA ? 0 : OPB) + (PL == 1 ? sub * 4096 + mc * 1024 : sub * 3072 + plane * 1024);
            }
    } else if constexpr (PL == 3) {
        // chunk = 32 rows x 32 B (the whole 16-k tile) of one plane; chunk c of a piece = row group c & 3, plane c >> 2
        const int srow = lane >> 1, sslot = (lane & 1) ^ ((lane >> 5) & 1);     // source k-slot: swapped for rows 16-31 (lane >> 5 = srow >> 4)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = wave + 8 * i, rg = c & 3, plane = c >> 2;
                int row;
                if (p == 0)      row = (rg >> 1) * 128 + (rg & 1) * 32;
                else if (p == 3) row = (rg >> 1) * 128 + 64 + (rg & 1) * 32;
                else if (p == 1) row = rg * 64;
                else             row = rg * 64 + 32;
                const bool isA = (p == 0 || p == 3);
                const long long r0 = isA ? m0 : n0, rmax = isA ? g.M : g.N, ld = isA ? g.lda : g.ldb;
                long long rgl = r0 + row + srow; if (rgl > rmax - 1) rgl = rmax - 1;
                voff[p][i] = (unsigned)(((rgl - r0) * ld + sslot * 8) * 2 + plane * (isA ? g.planeA : g.planeB));
                dst[p][i] = lds_base + (isA ? 0 : OPB) + plane * 8192 + row * 32;
            }
    } else {
        const int kh = wave & 1;
        const int srow = lane >> 2, sslot = (lane & 3) ^ ((lane >> 4) & 3);     // source k-slot of the lane's 16 bytes (XOR swizzle, see gemm.hip dma_offset)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rg = (wave >> 1) + 4 * i;                              // 0..7: sixteen-row group inside the piece
                int row;                                                         // tile row of the chunk's first row
                if (p == 0)      row = (rg >> 2) * 128 + (rg & 3) * 16;
                else if (p == 3) row = (rg >> 2) * 128 + 64 + (rg & 3) * 16;
                else if (p == 1) row = (rg >> 1) * 64 + (rg & 1) * 16;
                else             row = (rg >> 1) * 64 + 32 + (rg & 1) * 16;
                const bool isA = (p == 0 || p == 3);
                const long long r0 = isA ? m0 : n0, rmax = isA ? g.M : g.N, ld = isA ? g.lda : g.ldb;
                long long rgl = r0 + row + srow; if (rgl > rmax - 1) rgl = rmax - 1;
                voff[p][i] = (unsigned)(((rgl - r0) * ld + kh * 32 + sslot * 8) * 2);
                dst[p][i] = lds_base + (isA ? 0 : OP_BYTES) + kh * KHALF_BYTES + row * 64;
            }
    }
    const char* baseA = (const char*)(WG ? g.A + k_begin * g.lda + m0 : g.A + m0 * g.lda);
    const char* baseB = (const char*)(WG ? g.B + k_begin * g.ldb + n0 : g.B + n0 * g.ldb);
    const long long stepA = WG ? (long long)BK_ * 2 * g.lda : (long long)BK_ * 2, stepB = WG ? (long long)BK_ * 2 * g.ldb : (long long)BK_ * 2;

#define P_ISSUE(piece, stage_off)                                                                                   \
    do {                                                                                                            \
        const char* sb_ = ((piece) == 0 || (piece) == 3) ? baseA : baseB;                                           \
        p_glds16(voff[piece][0], sb_, dst[piece][0] + (stage_off));                                                 \
        if (second) p_glds16(voff[piece][1], sb_, dst[piece][1] + (stage_off));                                     \
    } while (0)

    // ---- fragment read offsets inside an operand image: lane (l31, h) reads row l31 of a 32-row sub-tile, 16-byte slot (2 jj + h) of k-half kh
    unsigned fa_off[2], fb_off[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const unsigned sl = (unsigned)(((2 * jj + h) ^ ((l31 >> 2) & 3)) * 16);
        fa_off[jj] = (unsigned)((wr * 128 + l31) * 64) + sl;
        fb_off[jj] = (unsigned)OPB + (unsigned)((wc * 64 + l31) * 64) + sl;
    }
    // PL 3: row l31 of a 32-row group, k-slot h (swapped for rows 16-31); + 1024 per row group, + 8192 per plane
    const unsigned fa3_off = (unsigned)((wr * 128 + l31) * 32 + ((h ^ ((l31 >> 4) & 1)) * 16));
    const unsigned fb3_off = (unsigned)OPB + (unsigned)((wc * 64 + l31) * 32 + ((h ^ ((l31 >> 4) & 1)) * 16));
    // WG: lane (group g4 = lane >> 4, i = lane & 15) reads k-row 8 (g4 >> 1) + (i >> 2) [+ 4 for the second read], columns 16 (g4 & 1) + 4 (i & 3) .. + 3
    // of its 32-column sub-run; + 1024 per 16-k step j, + 4096 per sub-run
    const unsigned tr_lane = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + 8 * (lane & 3));
    const unsigned tra_off = tr_lane + (unsigned)(wr * 4 * SUBB), trb_off = (unsigned)OPB + tr_lane + (unsigned)(wc * 2 * SUBB);
    // bias gradient (WG): row sums of A^T from the fragments the MFMAs consume; the four wave columns of a row half read the SAME A
    // fragments, so wave column wc sums the 16-k step j == wc only (a quarter of the VALU work each), combined through LDS at the end
    const bool do_rowsum = WG && g.rowsum != nullptr && tile_n == 0;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: the whole first k-tile, in the order of first use; its first two pieces have landed before anyone reads
    P_ISSUE(0, 0); P_ISSUE(1, 0); P_ISSUE(2, 0); P_ISSUE(3, 0);
    baseA += stepA; baseB += stepB;
    if (second) p_wait_vmcnt<4>(); else p_wait_vmcnt<2>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave half runs one barrier behind the first

#ifdef DLRM_TUNING
    const int P_DBG = g.debug;          // timing-only ablation switches (profiles/round4/bf16_gemm_ablation.txt)
#else
    constexpr int P_DBG = 0;
#endif
    bool first_tile = true;
    uintx4 fa[2][4], fb0[4], fb1[4];                    // fragments: A [2 sub-tile rows][4 k-steps] of the current row half, B column 0 / 1 [4 k-steps]
    unsigned cur = 0;                                   // byte offset of the stage being multiplied
#define P_READ_A(half)                                                                                              \
        if (!(P_DBG & 2) || first_tile)                                                                             \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                            \
            if constexpr (WG) fa[t][j] = p_tr_read8(lds_base + cur + tra_off + ((half) * 2 + t) * SUBB + j * 1024); \
            else if constexpr (PL == 3) fa[t][j] = *(const uintx4*)(ldsb + cur + fa3_off + j * 8192 + ((half) * 2 + t) * 1024); \
            else fa[t][j] = *(const uintx4*)(ldsb + cur + fa_off[j & 1] + (j >> 1) * KHALF_BYTES + ((half) * 2 + t) * 2048); \
        }
#define P_READ_B(FB, tn)                                                                                            \
        if (!(P_DBG & 2) || first_tile)                                                                             \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                            \
            if constexpr (WG) FB[j] = p_tr_read8(lds_base + cur + trb_off + (tn) * SUBB + j * 1024);                \
            else if constexpr (PL == 3) FB[j] = *(const uintx4*)(ldsb + cur + fb3_off + j * 8192 + (tn) * 1024);    \
            else FB[j] = *(const uintx4*)(ldsb + cur + fb_off[j & 1] + (j >> 1) * KHALF_BYTES + (tn) * 2048);        \
        }
    // (bf16 -> fp32 is a 16-bit shift: two VALU per packed pair)
#define P_ROWSUM(half)                                                                                              \
        if constexpr (WG) { if (do_rowsum && (PL == 1 || wc < 3)) {      /* PL 3: wave column wc sums plane wc (h + m + l == x) */ \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                         \
                const uintx4 f_ = wc == 0 ? fa[t][0] : wc == 1 ? fa[t][1] : wc == 2 ? fa[t][2] : fa[t][NJ - 1];     \
                float a_ = 0.f;                                                                                     \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
                    a_ += __uint_as_float(f_[e] << 16) + __uint_as_float(f_[e] & 0xffff0000u);                      \
                rs[(half) * 2 + t] += a_;                                                                           \
            } } }
    // MATH: the quadrant's 8 MFMAs at raised priority, with the two LDS-DMA issues of ONE piece of the next k-tile between them (an
    // LDS-DMA issue costs ~60 cycles among bare MFMAs but 100-185 in a segment that also carries the fragment reads — MI355X_MICROARCH.md;
    // round 4's first version issued them in the LOAD segment and ran 534 instead of ~300 cycles per half-phase).  sched_barrier pins the order.
#define P_MM(half, tn, FB, j, t) if (!(P_DBG & 4)) acc[(half) * 2 + (t)][tn] = P_MFMA(FB[j], fa[t][j], acc[(half) * 2 + (t)][tn]);
    // PL 3: product of B plane pb with A plane pa (planes 0 / 1 / 2 = h / m / l)
#define P_MM3(half, tn, FB, pb, pa, t) if (!(P_DBG & 4)) acc[(half) * 2 + (t)][tn] = P_MFMA(FB[pb], fa[t][pa], acc[(half) * 2 + (t)][tn]);
#define P_PIN(half, tn) asm volatile("" : "+v"(acc[(half) * 2][tn]), "+v"(acc[(half) * 2 + 1][tn]));
#define P_DMA(piece, i, MORE)                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (MORE && !(P_DBG & 1) && ((i) == 0 || second))                                                           \
            p_glds16(voff[piece][i], ((piece) == 0 || (piece) == 3) ? baseA : baseB, dst[piece][i] + nxt);          \
        __builtin_amdgcn_sched_barrier(0);
#define P_MATH(half, tn, FB, piece, MORE)                                                                           \
        __builtin_amdgcn_s_barrier();                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                              \
        if constexpr (PL == 1) {                                                                                    \
            P_MM(half, tn, FB, 0, 0) P_MM(half, tn, FB, 0, 1)                                                       \
            P_PIN(half, tn)                                                                                         \
            P_DMA(piece, 0, MORE)                                                                                   \
            P_MM(half, tn, FB, 1, 0) P_MM(half, tn, FB, 1, 1) P_MM(half, tn, FB, 2, 0)                              \
            P_PIN(half, tn)                                                                                         \
            P_DMA(piece, 1, MORE)                                                                                   \
            P_MM(half, tn, FB, 2, 1) P_MM(half, tn, FB, 3, 0) P_MM(half, tn, FB, 3, 1)                              \
        } else {                                                                                                    \
            /* six products per accumulator, smallest terms first, in the order of gemm3_kernel<ARITH = 1> (gemm.hip GEMM3_PRODUCT) */ \
            P_MM3(half, tn, FB, 2, 0, 0) P_MM3(half, tn, FB, 2, 0, 1)                                               \
            P_PIN(half, tn)                                                                                         \
            P_DMA(piece, 0, MORE)                                                                                   \
            P_MM3(half, tn, FB, 0, 2, 0) P_MM3(half, tn, FB, 0, 2, 1) P_MM3(half, tn, FB, 1, 1, 0) P_MM3(half, tn, FB, 1, 1, 1) \
            P_PIN(half, tn)                                                                                         \
            P_DMA(piece, 1, MORE)                                                                                   \
            P_MM3(half, tn, FB, 1, 0, 0) P_MM3(half, tn, FB, 1, 0, 1) P_MM3(half, tn, FB, 0, 1, 0) P_MM3(half, tn, FB, 0, 1, 1) \
            P_MM3(half, tn, FB, 0, 0, 0) P_MM3(half, tn, FB, 0, 0, 1)                                               \
        }                                                                                                           \
        /* (MFMAs are pure register operations: the IR optimizer SINKS them towards their next use, across setprio and the barrier, \
           into the next segment — an empty volatile asm that claims to rewrite the two accumulators pins them here) */             \
        P_PIN(half, tn)                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        __builtin_amdgcn_s_barrier();                                                                               \
        asm volatile("" ::: "memory");
    // LOAD ends with the counted wait: MORE = 1 leaves ONE piece (two DMA) in flight across the barrier — the piece issued in the MATH
    // segment before this LOAD; everything older has landed, so what the NEXT phase reads is complete for every wave after the barrier.
    // The last k-tile (MORE = 0, peeled) only drains.
#define P_WAIT(MORE) if (MORE) { if (second) p_wait_vmcnt<2>(); else p_wait_vmcnt<1>(); } else p_wait_vmcnt<0>();
#define P_KTILE(MORE)                                                                                               \
    {                                                                                                               \
        const unsigned nxt = cur ^ (unsigned)STG;                                                                   \
        /* phase 1: quadrant (rows 0-63, cols 0-31); piece A0 of the next k-tile */                                \
        P_READ_A(0) P_READ_B(fb0, 0) P_WAIT(MORE) P_MATH(0, 0, fb0, 0, MORE) P_ROWSUM(0)                            \
        /* phase 2: (rows 0-63, cols 32-63); piece B0 */                                                           \
        P_READ_B(fb1, 1) P_WAIT(MORE) P_MATH(0, 1, fb1, 1, MORE)                                                    \
        /* phase 3: (rows 64-127, cols 32-63); piece B1 */                                                         \
        P_READ_A(1) P_WAIT(MORE) P_MATH(1, 1, fb1, 2, MORE) P_ROWSUM(1)                                             \
        /* phase 4: (rows 64-127, cols 0-31): both operands are still in registers; piece A1 */                    \
        P_WAIT(MORE) P_MATH(1, 0, fb0, 3, MORE)                                                                     \
        if (MORE) { baseA += stepA; baseB += stepB; }                                                               \
        cur = nxt; first_tile = false;                                                                              \
    }
    for (int kt = 0; kt + 1 < nk; ++kt) P_KTILE(1)
    P_KTILE(0)
#undef P_KTILE
#undef P_READ_A
#undef P_READ_B
#undef P_ROWSUM
#undef P_MATH
#undef P_MM
#undef P_MM3
#undef P_PIN
#undef P_DMA
#undef P_WAIT
#undef P_ISSUE
    if (wr == 0) __builtin_amdgcn_s_barrier();          // the first half waits for the second: the tile buffers become epilogue staging
    __builtin_amdgcn_s_barrier();
#ifdef DLRM_TUNING
    if (P_DBG & 8) {                                    // timing only: NO epilogue (what prologue + k-loop cost without the band staging and the stores)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_ += acc[i][j][r];
        if (s_ == 123.456f && g.C) g.C[0] = s_;         // keeps the accumulators live
        return;
    }
#endif

    if constexpr (WG) {
        if (do_rowsum) {
            // lanes l31 / l31 + 32 hold different k of the same row; the four wave columns hold different 16-k steps: summed in a fixed order
            float* P = lds + 8 * (32 * P_EPI_LD);                   // behind the eight waves' staging areas: [wave][4 bands][32 rows]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = rs[t] + __shfl_xor(rs[t], 32, 64);
                if (lane < 32) P[wave * 128 + t * 32 + lane] = v;
            }
        }
        __builtin_amdgcn_s_barrier();
        if (do_rowsum && wc == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (lane < 32) {
                    const float* P = lds + 8 * (32 * P_EPI_LD) + t * 32 + lane;
                    const float v = ((P[(wr * 4 + 0) * 128] + P[(wr * 4 + 1) * 128]) + P[(wr * 4 + 2) * 128]) + P[(wr * 4 + 3) * 128];
                    const long long m = m0 + wr * 128 + t * 32 + lane;
                    if (m < g.M) g.rowsum[(long long)zs * g.M + m] = v;
                }
            }
        }
    }
    float* const Cz = WG && g.C ? g.C + (long long)zs * g.c_split_stride : g.C;
    // ---- epilogue (gemm3_kernel's, for a 128 x 64 wave tile): one 32-row band at a time through wave-private LDS.
    // Transposed C/D layout of the 32x32 MFMA: lane owns m_local = lane & 31 and n_local = 8 q + 4 (lane >> 5) + {0..3}, q = reg >> 2.
    const int c4 = (lane & 15) * 4;
    const long long nb = n0 + wc * 64 + c4;
    unsigned mkb[4] = {0u, 0u, 0u, 0u};
    if (g.bits_in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long mb = (m0 + wr * 128 + i * 32) >> 5, nbk = (n0 + wc * 64) >> 6, last_band = (g.M - 1) >> 5;
            mkb[i] = (nbk < g.bits_nblk) ? g.bits_in[((mb < last_band ? mb : last_band) * g.bits_nblk + nbk) * 64 + lane] : 0u;
        }
    }
    float* S = lds + wave * (32 * P_EPI_LD);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && nb < g.N) bv = *(const float4*)(g.bias + nb);      // N % 4 == 0 and nb % 4 == 0: the quad is inside the bias vector
    if constexpr (EP == 3) {
        // weight gradient: the fp32 slab of this k-slice, plain 16-byte row segments
        auto band_wg = [&](auto TMC) {
            constexpr int tm = decltype(TMC)::value;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                    *(float4*)__builtin_assume_aligned(S + l31 * P_EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const long long m = m0 + wr * 128 + tm * 32 + row;
                const float4 v = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c4, 16);
                if (m < g.M && nb < g.N) *(float4*)(Cz + m * g.ldc + nb) = v;
            }
        };
        band_wg(IC<0>{}); band_wg(IC<1>{}); band_wg(IC<2>{}); band_wg(IC<3>{});
        return;
    }
    if constexpr (EP == 1 || EP == 2) {
        // bf16-only output (host: wide16 preconditions hold, no addend / mul / Ub / fp32 output, act none | ReLU)
        const int c8 = (lane & 7) * 8;
        const long long nb8 = n0 + wc * 64 + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (g.bias && nb8 < g.N) { b0 = *(const float4*)(g.bias + nb8); b1 = *(const float4*)(g.bias + nb8 + 4); }
        asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w), "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w),
                     "+v"(mkb[0]), "+v"(mkb[1]), "+v"(mkb[2]), "+v"(mkb[3]));      // every load of the epilogue is waited for HERE, before the first store
        auto relu = [](float x) { return EP == 2 ? (x > 0.f ? x : 0.f) : x; };
        auto band_fast = [&](auto TMC) {
            constexpr int tm = decltype(TMC)::value;
            unsigned myword = 0u;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                    *(float4*)__builtin_assume_aligned(S + l31 * P_EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
                }
            if (g.bits_out) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float4 v = *(const float4*)__builtin_assume_aligned(S + (it * 4 + (lane >> 4)) * P_EPI_LD + c4, 16);
                    const float x0 = v.x + bv.x, x1 = v.y + bv.y, x2 = v.z + bv.z, x3 = v.w + bv.w;          // (ReLU(x) > 0 <=> x > 0)
                    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                 : "+v"(myword) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                const long long m = m0 + wr * 128 + tm * 32 + row;
                float4 v0 = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c8, 16);
                float4 v1 = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c8 + 4, 16);
                v0.x = relu(v0.x + b0.x); v0.y = relu(v0.y + b0.y); v0.z = relu(v0.z + b0.z); v0.w = relu(v0.w + b0.w);
                v1.x = relu(v1.x + b1.x); v1.y = relu(v1.y + b1.y); v1.z = relu(v1.z + b1.z); v1.w = relu(v1.w + b1.w);
                if (g.bits_in) {
                    const int src = (row & 3) * 16 + (lane & 7) * 2;
                    const unsigned w0 = (unsigned)__shfl((int)mkb[tm], src, 64), w1 = (unsigned)__shfl((int)mkb[tm], src + 1, 64);
                    const int sh = 31 - 4 * (row >> 2);
                    if (!((w0 >> (sh - 0)) & 1u)) v0.x = 0.f;
                    if (!((w0 >> (sh - 1)) & 1u)) v0.y = 0.f;
                    if (!((w0 >> (sh - 2)) & 1u)) v0.z = 0.f;
                    if (!((w0 >> (sh - 3)) & 1u)) v0.w = 0.f;
                    if (!((w1 >> (sh - 0)) & 1u)) v1.x = 0.f;
                    if (!((w1 >> (sh - 1)) & 1u)) v1.y = 0.f;
                    if (!((w1 >> (sh - 2)) & 1u)) v1.z = 0.f;
                    if (!((w1 >> (sh - 3)) & 1u)) v1.w = 0.f;
                }
                if (m < g.M && nb8 < g.N) {
                    if constexpr (PL == 1) {
                        uintx4 pk;
                        pk[0] = p_cvt_pk_bf16(v0.x, v0.y); pk[1] = p_cvt_pk_bf16(v0.z, v0.w);
                        pk[2] = p_cvt_pk_bf16(v1.x, v1.y); pk[3] = p_cvt_pk_bf16(v1.z, v1.w);
                        *(uintx4*)(g.Cb + m * g.ldcb + nb8) = pk;
                    } else {
                        uintx4 ph, pm, pl;
                        { unsigned h_, m_, l_; p_split2(v0.x, v0.y, h_, m_, l_); ph[0] = h_; pm[0] = m_; pl[0] = l_; } { unsigned h_, m_, l_; p_split2(v0.z, v0.w, h_, m_, l_); ph[1] = h_; pm[1] = m_; pl[1] = l_; }
                        { unsigned h_, m_, l_; p_split2(v1.x, v1.y, h_, m_, l_); ph[2] = h_; pm[2] = m_; pl[2] = l_; } { unsigned h_, m_, l_; p_split2(v1.z, v1.w, h_, m_, l_); ph[3] = h_; pm[3] = m_; pl[3] = l_; }
                        unsigned short* cb = g.Cb + m * g.ldcb + nb8;
                        *(uintx4*)cb = ph; *(uintx4*)(cb + g.planeC) = pm; *(uintx4*)(cb + 2 * g.planeC) = pl;
                    }
                }
            }
            if (g.bits_out) {
                const long long mb = (m0 + wr * 128 + tm * 32) >> 5, nbk = (n0 + wc * 64) >> 6;
                if (mb <= ((g.M - 1) >> 5) && nbk < g.bits_nblk) g.bits_out[(mb * g.bits_nblk + nbk) * 64 + lane] = myword;
            }
        };
        band_fast(IC<0>{}); band_fast(IC<1>{}); band_fast(IC<2>{}); band_fast(IC<3>{});
        return;
    }
    // (one instantiation per band through a generic lambda: with the second output form the unroller refused the pragma on a plain loop —
    // "unrolled size is too large" — and a rolled loop indexes acc[tm] dynamically, i.e. puts the accumulators into scratch memory)
    auto band = [&](auto TMC) {
        constexpr int tm = decltype(TMC)::value;
        unsigned myword = 0u;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
                *(float4*)__builtin_assume_aligned(S + l31 * P_EPI_LD + tn * 32 + 8 * q + 4 * h, 16) = v;
            }
        if (!WG && g.wide16) {
            // bf16-ONLY output (lean storage): the band leaves as 16-byte stores — a lane owns 8 consecutive columns of one row, four store
            // instructions per band instead of eight 8-byte ones (the epilogue is store-ISSUE-bound, ~7 B/clk/CU: MI355X_MICROARCH.md).
            // The ReLU sign bits keep their documented ownership (dlrm_relu_bits_bytes: lane l = rows it*4 + (l >> 4), columns 4 (l & 15) + c):
            // written from a separate pass over the staged band in that ownership, read through two cross-lane fetches per row.
            if (g.bits_out) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float4 v = *(const float4*)__builtin_assume_aligned(S + (it * 4 + (lane >> 4)) * P_EPI_LD + c4, 16);
                    const float x0 = v.x + bv.x, x1 = v.y + bv.y, x2 = v.z + bv.z, x3 = v.w + bv.w;          // (ReLU(x) > 0 <=> x > 0)
                    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                                 "v_cmp_lt_f32 vcc, 0, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                 : "+v"(myword) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
                }
            }
            const int c8 = (lane & 7) * 8;
            const long long nb8 = n0 + wc * 64 + c8;
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (g.bias && nb8 < g.N) { b0 = *(const float4*)(g.bias + nb8); b1 = *(const float4*)(g.bias + nb8 + 4); }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                const long long m = m0 + wr * 128 + tm * 32 + row;
                float4 v0 = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c8, 16);
                float4 v1 = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c8 + 4, 16);
                v0.x = p_act(v0.x + b0.x, g.act); v0.y = p_act(v0.y + b0.y, g.act); v0.z = p_act(v0.z + b0.z, g.act); v0.w = p_act(v0.w + b0.w, g.act);
                v1.x = p_act(v1.x + b1.x, g.act); v1.y = p_act(v1.y + b1.y, g.act); v1.z = p_act(v1.z + b1.z, g.act); v1.w = p_act(v1.w + b1.w, g.act);
                if (g.bits_in) {
                    // the word of the lane that owns (row, column quad) in the documented layout: lane (row & 3) * 16 + quad, bit 31 - (4 (row >> 2) + c)
                    const int src = (row & 3) * 16 + (lane & 7) * 2;
                    const unsigned w0 = (unsigned)__shfl((int)mkb[tm], src, 64), w1 = (unsigned)__shfl((int)mkb[tm], src + 1, 64);
                    const int sh = 31 - 4 * (row >> 2);
                    if (!((w0 >> (sh - 0)) & 1u)) v0.x = 0.f;
                    if (!((w0 >> (sh - 1)) & 1u)) v0.y = 0.f;
                    if (!((w0 >> (sh - 2)) & 1u)) v0.z = 0.f;
                    if (!((w0 >> (sh - 3)) & 1u)) v0.w = 0.f;
                    if (!((w1 >> (sh - 0)) & 1u)) v1.x = 0.f;
                    if (!((w1 >> (sh - 1)) & 1u)) v1.y = 0.f;
                    if (!((w1 >> (sh - 2)) & 1u)) v1.z = 0.f;
                    if (!((w1 >> (sh - 3)) & 1u)) v1.w = 0.f;
                }
                if (m < g.M && nb8 < g.N && !(P_DBG & 16)) {        // (16: timing only — the whole epilogue EXCEPT its global stores)
                    if constexpr (PL == 1) {
                        uintx4 pk;
                        pk[0] = p_cvt_pk_bf16(v0.x, v0.y); pk[1] = p_cvt_pk_bf16(v0.z, v0.w);
                        pk[2] = p_cvt_pk_bf16(v1.x, v1.y); pk[3] = p_cvt_pk_bf16(v1.z, v1.w);
                        *(uintx4*)(g.Cb + m * g.ldcb + nb8) = pk;
                    } else {
                        uintx4 ph, pm, pl;
                        { unsigned h_, m_, l_; p_split2(v0.x, v0.y, h_, m_, l_); ph[0] = h_; pm[0] = m_; pl[0] = l_; } { unsigned h_, m_, l_; p_split2(v0.z, v0.w, h_, m_, l_); ph[1] = h_; pm[1] = m_; pl[1] = l_; }
                        { unsigned h_, m_, l_; p_split2(v1.x, v1.y, h_, m_, l_); ph[2] = h_; pm[2] = m_; pl[2] = l_; } { unsigned h_, m_, l_; p_split2(v1.z, v1.w, h_, m_, l_); ph[3] = h_; pm[3] = m_; pl[3] = l_; }
                        unsigned short* cb = g.Cb + m * g.ldcb + nb8;
                        *(uintx4*)cb = ph; *(uintx4*)(cb + g.planeC) = pm; *(uintx4*)(cb + 2 * g.planeC) = pl;
                    }
                }
            }
        } else
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + (lane >> 4);
            const long long m = m0 + wr * 128 + tm * 32 + row;
            const bool live = m < g.M && nb < g.N;
            float4 v = *(const float4*)__builtin_assume_aligned(S + row * P_EPI_LD + c4, 16);
            v.x = p_act(v.x + bv.x, g.act); v.y = p_act(v.y + bv.y, g.act);
            v.z = p_act(v.z + bv.z, g.act); v.w = p_act(v.w + bv.w, g.act);
            if (g.bits_out) {           // word = 2*word + (v > 0): compare into VCC, add-with-carry (layout: dlrm_relu_bits_bytes, dlrm_hip.h)
                asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                             "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                             "v_cmp_lt_f32 vcc, 0, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                             "v_cmp_lt_f32 vcc, 0, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                             : "+v"(myword) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "vcc");
            }
            if (!live || (P_DBG & 16)) continue;
            if (g.bits_in) {
                const unsigned wv = mkb[tm];
                if (!((wv >> (31 - (it * 4 + 0))) & 1u)) v.x = 0.f;
                if (!((wv >> (31 - (it * 4 + 1))) & 1u)) v.y = 0.f;
                if (!((wv >> (31 - (it * 4 + 2))) & 1u)) v.z = 0.f;
                if (!((wv >> (31 - (it * 4 + 3))) & 1u)) v.w = 0.f;
            }
            if constexpr (PL == 1) {
                if (g.Ub) {                  // the cross layer's u, as the backward pass reads it
                    uint2 pk; pk.x = p_cvt_pk_bf16(v.x, v.y); pk.y = p_cvt_pk_bf16(v.z, v.w);
                    *(uint2*)(g.Ub + m * g.ldub + nb) = pk;
                }
            }
            if (g.mul) {                     // x_{l+1} = fma(x0, u, xl)
                const float4 x0 = *(const float4*)(g.mul + m * g.ldmul + nb);
                const float4 a = *(const float4*)(g.addend + m * g.ldadd + nb);
                v.x = __builtin_fmaf(x0.x, v.x, a.x); v.y = __builtin_fmaf(x0.y, v.y, a.y);
                v.z = __builtin_fmaf(x0.z, v.z, a.z); v.w = __builtin_fmaf(x0.w, v.w, a.w);
            } else if (g.addend) {
                const float4 a = *(const float4*)(g.addend + m * g.ldadd + nb);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            if (g.addend2) {
                const float4 a = *(const float4*)(g.addend2 + m * g.ldadd2 + nb);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            if (Cz) *(float4*)(Cz + m * g.ldc + nb) = v;
            if (g.Cb) {
                if constexpr (PL == 1) {
                    uint2 pk; pk.x = p_cvt_pk_bf16(v.x, v.y); pk.y = p_cvt_pk_bf16(v.z, v.w);
                    *(uint2*)(g.Cb + m * g.ldcb + nb) = pk;
                } else {
                    uint2 ph, pm, pl;
                    { unsigned h_, m_, l_; p_split2(v.x, v.y, h_, m_, l_); ph.x = h_; pm.x = m_; pl.x = l_; } { unsigned h_, m_, l_; p_split2(v.z, v.w, h_, m_, l_); ph.y = h_; pm.y = m_; pl.y = l_; }
                    unsigned short* cb = g.Cb + m * g.ldcb + nb;
                    *(uint2*)cb = ph; *(uint2*)(cb + g.planeC) = pm; *(uint2*)(cb + 2 * g.planeC) = pl;
                }
            }
        }
        if (g.bits_out) {
            const long long mb = (m0 + wr * 128 + tm * 32) >> 5, nbk = (n0 + wc * 64) >> 6;
            if (mb <= ((g.M - 1) >> 5) && nbk < g.bits_nblk) g.bits_out[(mb * g.bits_nblk + nbk) * 64 + lane] = myword;
        }
    };
    band(IC<0>{}); band(IC<1>{}); band(IC<2>{}); band(IC<3>{});
}

}  // namespace

constexpr int LDS_PL1 = 2 * STAGE_BYTES;              // 128 KiB: one workgroup per CU
constexpr int LDS_PL3 = 2 * 2 * (3 * 256 * 32);       //  96 KiB (>= the 72 KiB of epilogue staging + bias-gradient partials)
static void phased_attr(const void* fn, bool& done, int bytes = LDS_PL1) {
    if (!done) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done = true;
    }
}

static int phased_enabled() {           // tuning builds, DLRM_BF16_PHASED=0: keep the fp32-shaped kernels everywhere (A/B runs)
    static const int enabled = DLRM_TUNE_ENV("DLRM_BF16_PHASED", 1);
    return enabled;
}

// returns 0 when the phased kernel took the call, DLRM_GEMV_NOT_HANDLED when the shape is outside its preconditions (the caller keeps gemm3_kernel)
int dlrm_gemm_bf16_phased(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias, int act,
                          uint64_t* relu_bits_out, const uint64_t* relu_bits_in, const float* addend, int64_t ldadd, const float* addend2,
                          int64_t ldadd2, float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, hipStream_t st, const float* mul, int64_t ldmul,
                          uint16_t* Ub, int64_t ldub) {
    if (!phased_enabled() || K % PBK || N % 4 || N < 192 || M < 256 || lda % 8 || ldb % 8) return DLRM_GEMV_NOT_HANDLED;
    if (bias && !dlrm_aligned16(bias)) return DLRM_GEMV_NOT_HANDLED;
    BfArgs g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.Cb = Cb; g.ldcb = ldcb;
    g.bias = bias; g.act = act; g.addend = addend; g.ldadd = ldadd; g.addend2 = addend2; g.ldadd2 = ldadd2;
    g.mul = mul; g.ldmul = ldmul; g.Ub = Ub; g.ldub = ldub;
    g.bits_out = (unsigned*)relu_bits_out; g.bits_in = (const unsigned*)relu_bits_in; g.bits_nblk = ((long long)N + 63) / 64;
    g.tiles_m = (int)((M + PBM - 1) / PBM); g.tiles_n = (int)((N + PBN - 1) / PBN);
    static const int wide = DLRM_TUNE_ENV("DLRM_BF16_WIDE_STORE", 1);      // tuning builds: 0 keeps the 8-byte bf16 stores
#ifdef DLRM_TUNING
    { static int dbg = -1; if (dbg < 0) dbg = DLRM_DEBUG_ENV("DLRM_BF16_DEBUG", 31); g.debug = dbg; }
#endif
    g.wide16 = (wide && Cb && !C && !addend && !mul && !Ub && N % 8 == 0 && ldcb % 8 == 0 && dlrm_aligned16(Cb) && (!bias || dlrm_aligned16(bias))) ? 1 : 0;
    // (a DIRECT epilogue — bias / activation / rounding in registers, v_permlane32_swap to 16 contiguous bytes per lane, no LDS round trip — was
    // built and measured in round 5: correct, 4-6 % slower than this staged one; profiles/round5/bf16_epilogue_ablation.txt)
    // the straight-line epilogue where the call is bf16-only with no / ReLU activation (the lean towers' hidden layers); tuning builds: DLRM_BF16_EPI=0
    static const int epi_on = DLRM_TUNE_ENV("DLRM_BF16_EPI", 1);
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    if (epi_on && g.wide16 && (act == DLRM_ACT_NONE || act == DLRM_ACT_RELU)) {
        static bool attr1_done[DLRM_MAX_DEVICES] = {}, attr2_done[DLRM_MAX_DEVICES] = {};
        if (act == DLRM_ACT_RELU) {
            phased_attr((const void*)gemm_bf16_phased_kernel<false, 1, 2>, attr2_done[dlrm_current_device()]);
            hipLaunchKernelGGL((gemm_bf16_phased_kernel<false, 1, 2>), grid, dim3(512), LDS_PL1, st, g);
        } else {
            phased_attr((const void*)gemm_bf16_phased_kernel<false, 1, 1>, attr1_done[dlrm_current_device()]);
            hipLaunchKernelGGL((gemm_bf16_phased_kernel<false, 1, 1>), grid, dim3(512), LDS_PL1, st, g);
        }
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    static bool attr_done[DLRM_MAX_DEVICES] = {};
    phased_attr((const void*)gemm_bf16_phased_kernel<false, 1>, attr_done[dlrm_current_device()]);
    hipLaunchKernelGGL((gemm_bf16_phased_kernel<false, 1>), grid, dim3(512), LDS_PL1, st, g);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// ---- "bf16x6" from pre-split planes (PL = 3).  No fallback kernel reads planes: the host mirror asks dlrm_gemm_bf16x6_ok first and keeps the
// fp32-storage path (gemm3_kernel<ARITH = 1>, which splits in its k-loop) for shapes outside these preconditions.
bool dlrm_gemm_bf16x6_ok(int64_t M, int N, int K, int64_t lda, int64_t ldb) {
    return phased_enabled() && K >= 16 && K % 16 == 0 && N % 4 == 0 && N >= 192 && M >= 256 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K;
}
// the planes of an operand travel in the 32-bit lane offset of the LDS-DMA: two plane strides + one tile of rows must stay below 4 GiB
static bool planes_reachable(int64_t plane_elems, int64_t ld, int64_t rows_in_flight) {
    return plane_elems >= 0 && plane_elems % 8 == 0 && (2 * plane_elems + rows_in_flight * ld + 64) * 2 < (int64_t)0xffffffffll;
}
int dlrm_gemm_bf16x6_phased(int64_t M, int N, int K, const uint16_t* A, int64_t lda, int64_t planeA, const uint16_t* B, int64_t ldb, int64_t planeB,
                            const float* bias, int act, uint64_t* relu_bits_out, const uint64_t* relu_bits_in, float* C, int64_t ldc, uint16_t* Cp,
                            int64_t ldcp, int64_t planeC, hipStream_t st) {
    if (!dlrm_gemm_bf16x6_ok(M, N, K, lda, ldb) || !planes_reachable(planeA, lda, 256) || !planes_reachable(planeB, ldb, 256)) return DLRM_E_ALIGN;
    if (bias && !dlrm_aligned16(bias)) return DLRM_E_ALIGN;
    if (Cp && (planeC % 4 || ldcp % 4)) return DLRM_E_ALIGN;
    BfArgs g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.Cb = Cp; g.ldcb = ldcp;
    g.planeA = planeA * 2; g.planeB = planeB * 2; g.planeC = planeC;
    g.bias = bias; g.act = act;
    g.bits_out = (unsigned*)relu_bits_out; g.bits_in = (const unsigned*)relu_bits_in; g.bits_nblk = ((long long)N + 63) / 64;
    g.tiles_m = (int)((M + PBM - 1) / PBM); g.tiles_n = (int)((N + PBN - 1) / PBN);
#ifdef DLRM_TUNING
    { static int dbg = -1; if (dbg < 0) dbg = DLRM_DEBUG_ENV("DLRM_BF16_DEBUG", 31); g.debug = dbg; }
#endif
    g.wide16 = (Cp && !C && N % 8 == 0 && ldcp % 8 == 0 && planeC % 8 == 0 && dlrm_aligned16(Cp)) ? 1 : 0;
    static bool attr_done[DLRM_MAX_DEVICES] = {};
    phased_attr((const void*)gemm_bf16_phased_kernel<false, 3>, attr_done[dlrm_current_device()], LDS_PL3);
    hipLaunchKernelGGL((gemm_bf16_phased_kernel<false, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(512), LDS_PL3, st, g);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// Weight-gradient form: slab z [N_out, ldc] (fp32) = sum over batch rows [z * kchunk, (z + 1) * kchunk) of dZ[r, n] * X[r, k]; rowsum z [N_out] =
// column sums of dZ over the same rows (nullable).  dZ [Mb, N_out] and X [Mb, K_in] bf16 row-major.  The caller sums the slabs.
bool dlrm_gemm_bf16_wgrad_ok(int64_t Mb, int N_out, int K_in, int64_t lddz, int64_t ldx) {
    return phased_enabled() && Mb >= 256 && Mb % PBK == 0 && N_out % 8 == 0 && K_in % 8 == 0 && N_out >= 64 && K_in >= 64 && lddz % 8 == 0 && ldx % 8 == 0;
}
void dlrm_gemm_bf16_wgrad_plan(int64_t Mb, int N_out, int K_in, int* splits_out, int64_t* kchunk_out) {
    const int tiles = ((N_out + PBM - 1) / PBM) * ((K_in + PBN - 1) / PBN);
    static int cu_count[DLRM_MAX_DEVICES] = {};                   // per device, queried once
    const int dev = dlrm_current_device();
    if (cu_count[dev] == 0) {
        int n = 0;
        cu_count[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    const int cus = cu_count[dev];
    // one workgroup per CU is resident: tiles x splits must not spill into a nearly empty second round (28 tiles x 10 slices = 280
    // workgroups ran 1.7x longer than 28 x 9 = 252); more than a round's worth of tiles: whole rounds
    int splits = tiles <= cus ? cus / tiles : 1;
    const int64_t max_splits = Mb / 256;                           // at least four k-tiles per slice
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    int64_t kchunk = (Mb + splits - 1) / splits;
    kchunk = ((kchunk + PBK - 1) / PBK) * PBK;
    *splits_out = (int)((Mb + kchunk - 1) / kchunk);
    *kchunk_out = kchunk;
}
// planes = 0: bf16 operands; planes = 1: each operand is three bf16 planes planeZ / planeX elements apart (bf16x6)
int dlrm_gemm_bf16_wgrad_phased(int64_t Mb, int N_out, int K_in, const uint16_t* dZ, int64_t lddz, const uint16_t* X, int64_t ldx,
                                float* slabs, int64_t ldc, int64_t slab_stride, float* rowsum_parts, int splits, int64_t kchunk, hipStream_t st,
                                int planes, int64_t planeZ, int64_t planeX) {
    if (!dlrm_gemm_bf16_wgrad_ok(Mb, N_out, K_in, lddz, ldx) || !dZ || !X || !slabs || ldc % 4 || !dlrm_aligned16(slabs) || !dlrm_aligned16(dZ) ||
        !dlrm_aligned16(X) || kchunk % PBK || splits < 1)
        return DLRM_E_ARG;
    BfArgs g = {};
    g.M = N_out; g.N = K_in; g.K = Mb;
    g.A = dZ; g.lda = lddz; g.B = X; g.ldb = ldx; g.C = slabs; g.ldc = ldc;
    g.act = DLRM_ACT_NONE;
    g.kchunk = kchunk; g.c_split_stride = slab_stride; g.rowsum = rowsum_parts;
    g.tiles_m = (N_out + PBM - 1) / PBM; g.tiles_n = (K_in + PBN - 1) / PBN;
    if (planes) {
        // (a k-tile of the k-strided form holds 16 batch rows of every plane)
        if (!planes_reachable(planeZ, lddz, 16) || !planes_reachable(planeX, ldx, 16)) return DLRM_E_ALIGN;
        g.planeA = planeZ * 2; g.planeB = planeX * 2;
        static bool attr3_done[DLRM_MAX_DEVICES] = {};
        phased_attr((const void*)gemm_bf16_phased_kernel<true, 3, 3>, attr3_done[dlrm_current_device()], LDS_PL3);
        hipLaunchKernelGGL((gemm_bf16_phased_kernel<true, 3, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits), dim3(512), LDS_PL3, st, g);
        DLRM_LAUNCH_CHECK();
        return 0;
    }
    static bool attr_done[DLRM_MAX_DEVICES] = {};       // (the weight gradient's slabs always leave through the straight-line fp32 epilogue, EP = 3)
    phased_attr((const void*)gemm_bf16_phased_kernel<true, 1, 3>, attr_done[dlrm_current_device()]);
    hipLaunchKernelGGL((gemm_bf16_phased_kernel<true, 1, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits), dim3(512), LDS_PL1, st, g);
    DLRM_LAUNCH_CHECK();
    return 0;
}
