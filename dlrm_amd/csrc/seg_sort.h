// seg_sort.h — the (table,row) sort of the sort-based embedding updates, written for what the lookups ARE instead of as a
// general 32-bit radix sort (rocPRIM's onesweep took 4 passes + a histogram + 9 hipMemsetAsync fills, ~150 of the
// 519 us of the fused EmbeddingBag backward + SGD, and cannot be replayed inside a HIP graph).
//
// Reference semantics served: the sparse gradient of EmbeddingBagBackward is consumed row by row IN INPUT ORDER (sparse SGD:
// `p.add_(coo, alpha=-lr)`, dlrm_s_pytorch.py:1613,1620; row-wise Adagrad coalesces, optim/rwsadagrad.py:117-120), so the update
// kernels want every table's lookups grouped by row with equal rows in input order: a STABLE sort by (table, row).
//
// What is special about the input:
//   * lookups arrive TABLE-MAJOR (positions base[t] .. base[t] + nnz[t]): the table bits of the key are already sorted, only the
//     row bits of each table's own segment need sorting, and a table with few rows needs few bits: the 13 Criteo-Terabyte tables
//     with <= 8192 rows take ONE counting pass, the 40 M-row tables two passes of 13 bits (rocPRIM: four passes of 8 over 31 bits);
//   * a segment is at most a few global batches long (one lookup per bag in the Criteo data sets).
//
// Algorithm: least-significant-digit radix sort, per table segment, digits of up to 13 bits, 1-3 rounds (tables that need fewer
// rounds join in the last ones; ping-pong buffers chosen so that every table ends in the OUT buffer).  One round = three launches
// over all participating tables, no memsets, no atomics on global memory, nothing that a HIP graph cannot replay:
//   seg_hist_kernel     one WAVE per tile of 2048 consecutive entries: digit histogram in LDS -> hist[table][tile][bin]
//   seg_colscan_kernel  one thread per (table, bin): exclusive prefix over the table's tiles (in place) + the bin's total
//   seg_binscan_kernel  one workgroup per table: exclusive prefix of the bins' totals = first position of every digit
//   seg_scatter_kernel  one wave per tile (its 2048 keys prefetched into registers, the start positions of ITS digits gathered), entries
//                       taken 64 at a time IN ORDER: lanes with equal digits find each other through a claim byte per digit + 6 ballots
//                       (match-any), the lowest lane of each group advances the digit's cursor in LDS, every entry goes to
//                       cursor + its rank in the group: stable by construction (rank order inside a 64-entry step, step order
//                       inside a tile, tile order through the prefix)
// Long segments (multi-hot batches: the 100-hot table of the MLPerf-v2 batch holds 6.5 M lookups = 3200 tiles): the per-tile histograms
// [tiles][bins] are kept inside SEG_HIST_BUDGET words per table by NARROWER digits (3200 tiles -> 11 bits -> three rounds over 26 row bits;
// rocPRIM's general sort took four 8-bit passes over all 31 key bits), and the prefix over the tiles is taken in GROUPS of SEG_GROUP_TILES
// tiles (seg_colscan_kernel per group, then seg_groupscan_kernel over the groups' totals) so that no thread walks more than 128 tiles.
#pragma once
#include "common.h"

namespace {

#ifndef DLRM_SEG_TILE
#define DLRM_SEG_TILE 2048
#endif
constexpr int SEG_TILE = DLRM_SEG_TILE;  // entries per wave-tile (measured at Criteo-Terabyte shapes: 4096 -> 134 us per sort, 2048 -> 120 us; -DDLRM_SEG_TILE: tools/build_variant_lib.sh)
constexpr int SEG_MAX_DBITS = 13;        // 8192 bins: 32 KB of LDS per wave
constexpr int SEG_GROUP_TILES = 128;     // tiles whose counts ONE thread of seg_colscan_kernel prefixes (262144 lookups); longer segments: several groups
constexpr int SEG_MAX_GROUPS = 256;      // per table segment (67 M lookups)
constexpr int SEG_MAX_ROUNDS = 4;
constexpr long long SEG_HIST_BUDGET = 8ll << 20;   // words of per-tile histogram per table and round (32 MB): bounds the digit width of long segments

struct SegRound {
    int ntab;                                                // participating tables of this round
    int tab[DLRM_MAX_TABLES_PER_LAUNCH];                     // their slot in the launch group
    unsigned tile_start[DLRM_MAX_TABLES_PER_LAUNCH + 1];     // prefix of their tile counts
    unsigned scan_start[DLRM_MAX_TABLES_PER_LAUNCH + 1];     // prefix of their 256-bin blocks (grid of seg_colscan_kernel)
    unsigned char dbits[DLRM_MAX_TABLES_PER_LAUNCH];         // digit width of this round
    unsigned char shift[DLRM_MAX_TABLES_PER_LAUNCH];         // digit position
    unsigned char first[DLRM_MAX_TABLES_PER_LAUNCH];         // the table's first round: source = IN, value = the position itself
    unsigned char dst_out[DLRM_MAX_TABLES_PER_LAUNCH];       // destination = OUT (else TMP); a later round's source is the other one
    unsigned hist_off[DLRM_MAX_TABLES_PER_LAUNCH];           // first counter of the table in the hist buffer  [tiles][bins]
    unsigned bin_off[DLRM_MAX_TABLES_PER_LAUNCH];            // first counter of the table in the bin-base buffer [bins]
    unsigned short ngroups[DLRM_MAX_TABLES_PER_LAUNCH];      // tile groups of the table (1: the bins' totals go straight to the bin-base buffer)
    unsigned gtot_off[DLRM_MAX_TABLES_PER_LAUNCH];           // first counter of the table in the group-total buffer [groups][bins]
    unsigned gscan_start[DLRM_MAX_TABLES_PER_LAUNCH + 1];    // prefix of the 256-bin blocks of the tables WITH groups (grid of seg_groupscan_kernel)
    long long base[DLRM_MAX_TABLES_PER_LAUNCH];              // first global position of the table's segment
    long long nnz[DLRM_MAX_TABLES_PER_LAUNCH];
};

struct SegPlan {
    int rounds;
    SegRound round[SEG_MAX_ROUNDS];
    size_t hist_words, bin_words, gtot_words;                // buffer sizes (u32 words), the maximum over rounds
};

static int seg_bits_for(long long n) { int b = 0; while (((long long)1 << b) < n) ++b; return b < 1 ? 1 : b; }

// false: a segment is too long for this sorter (or a table has too many row bits for SEG_MAX_ROUNDS digits) -> the caller uses the general sorter
// allow_long: take segments of more than SEG_GROUP_TILES tiles too.  Correct and tested (a 6.55 M-lookup segment), but MEASURED SLOWER than the
// general sorter there (profiles/round4/sort_long_segments.md: 14 M lookups of the MLPerf-v2 batch 1146 vs 773 us — three rounds of scattered
// 4-byte stores at 1-4 entries per bin and tile against onesweep's block-local reordering into coalesced runs), so the default keeps long
// segments on the general sorter and DLRM_SORT=own opts in.
static bool seg_plan(int n, const long long* nnz, const long long* rows, SegPlan* p, bool allow_long) {
    int passes[DLRM_MAX_TABLES_PER_LAUNCH], rb[DLRM_MAX_TABLES_PER_LAUNCH];
    int R = 0;
    for (int k = 0; k < n; ++k) {
        const long long tiles = (nnz[k] + SEG_TILE - 1) / SEG_TILE;
        if (tiles > (long long)SEG_GROUP_TILES * (allow_long ? SEG_MAX_GROUPS : 1)) return false;
        int dmax = SEG_MAX_DBITS;                             // widest digit whose [tiles][bins] histogram fits the budget
        while (dmax > 6 && (tiles << dmax) > SEG_HIST_BUDGET) --dmax;
        rb[k] = seg_bits_for(rows[k]);
        passes[k] = nnz[k] > 0 ? (rb[k] + dmax - 1) / dmax : 0;
        if (passes[k] > SEG_MAX_ROUNDS) return false;
        if (passes[k] > R) R = passes[k];
    }
    p->rounds = R;
    p->hist_words = 0; p->bin_words = 0; p->gtot_words = 0;
    long long base[DLRM_MAX_TABLES_PER_LAUNCH];
    long long acc = 0;
    for (int k = 0; k < n; ++k) { base[k] = acc; acc += nnz[k]; }
    for (int r = 0; r < R; ++r) {
        SegRound& q = p->round[r];
        q.ntab = 0; q.tile_start[0] = 0; q.scan_start[0] = 0; q.gscan_start[0] = 0;
        size_t hw = 0, bw = 0, gw = 0;
        for (int k = 0; k < n; ++k) {
            const int pass = r - (R - passes[k]);             // this table's pass index in round r
            if (passes[k] == 0 || pass < 0) continue;
            // digits as even as possible: the first (rb % passes) passes are one bit wider
            const int lo = rb[k] / passes[k], rem = rb[k] % passes[k];
            int shift = 0;
            for (int j = 0; j < pass; ++j) shift += lo + (j < rem ? 1 : 0);
            const int d = lo + (pass < rem ? 1 : 0);
            const int i = q.ntab++;
            const unsigned tiles = (unsigned)((nnz[k] + SEG_TILE - 1) / SEG_TILE);
            const unsigned groups = (tiles + SEG_GROUP_TILES - 1) / SEG_GROUP_TILES;
            const unsigned binblocks = (((unsigned)1 << d) + 255) / 256;
            q.tab[i] = k; q.tile_start[i + 1] = q.tile_start[i] + tiles;
            q.scan_start[i + 1] = q.scan_start[i] + binblocks * groups;
            q.gscan_start[i + 1] = q.gscan_start[i] + (groups > 1 ? binblocks : 0);
            q.dbits[i] = (unsigned char)d; q.shift[i] = (unsigned char)shift;
            q.first[i] = (unsigned char)(pass == 0);
            q.dst_out[i] = (unsigned char)(((R - 1 - r) & 1) == 0);
            q.hist_off[i] = (unsigned)hw; q.bin_off[i] = (unsigned)bw;
            q.ngroups[i] = (unsigned short)groups; q.gtot_off[i] = (unsigned)gw;
            q.base[i] = base[k]; q.nnz[i] = nnz[k];
            hw += (size_t)tiles << d; bw += (size_t)1 << d;
            if (groups > 1) gw += (size_t)groups << d;
        }
        for (int i = q.ntab; i < DLRM_MAX_TABLES_PER_LAUNCH; ++i) {
            q.tab[i] = 0; q.tile_start[i + 1] = q.tile_start[q.ntab]; q.scan_start[i + 1] = q.scan_start[q.ntab]; q.gscan_start[i + 1] = q.gscan_start[q.ntab];
            q.dbits[i] = 1; q.shift[i] = 0; q.first[i] = 0; q.dst_out[i] = 1;
            q.hist_off[i] = 0; q.bin_off[i] = 0; q.ngroups[i] = 1; q.gtot_off[i] = 0; q.base[i] = 0; q.nnz[i] = 0;
        }
        if (hw > p->hist_words) p->hist_words = hw;
        if (bw > p->bin_words) p->bin_words = bw;
        if (gw > p->gtot_words) p->gtot_words = gw;
        if (hw > 0xFFFFFFFFull || gw > 0xFFFFFFFFull) return false;      // (offsets are 32-bit words)
    }
    return true;
}

// which participating table owns wave-tile `w` (<= 32 entries: a scalar scan of the kernarg)
__device__ __forceinline__ int seg_find(const SegRound& q, unsigned w) {
    int i = 0;
    while (i + 1 < q.ntab && w >= q.tile_start[i + 1]) ++i;
    return i;
}

constexpr int SEG_CHUNKS = SEG_TILE / 64;     // 64-entry steps per tile: a lane keeps its SEG_CHUNKS keys of the tile in registers

// All loads of a tile are issued up front (SEG_CHUNKS independent, coalesced loads per lane): a wave that fetched one 64-entry step
// at a time paid a full memory round trip per step — 64 dependent round trips per tile, ~0.2 ms per pass on Criteo-Terabyte shapes
// (profiles/round3: the first version of this sorter was slower than rocPRIM for exactly that reason).
template <typename KT>
__global__ __launch_bounds__(64) void seg_hist_kernel(SegRound q, const KT* __restrict__ in, const KT* __restrict__ tmp,
                                                      const KT* __restrict__ out, unsigned* __restrict__ hist) {
    __shared__ unsigned h[1 << SEG_MAX_DBITS];
    const int lane = threadIdx.x;
    const int i = seg_find(q, blockIdx.x);
    const unsigned tile = blockIdx.x - q.tile_start[i];
    const int d = q.dbits[i], shift = q.shift[i];
    const unsigned bins = 1u << d, mask = bins - 1;
    const KT* __restrict__ src = q.first[i] ? in : (q.dst_out[i] ? tmp : out);
    const long long s = q.base[i] + (long long)tile * SEG_TILE;
    long long nn = q.nnz[i] - (long long)tile * SEG_TILE; if (nn > SEG_TILE) nn = SEG_TILE;
    const int n = (int)nn;
    KT k[SEG_CHUNKS];
    // (indices past the tile's end are clamped to its last entry, not predicated: 64 unconditional loads in straight-line code)
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j) { const int e = j * 64 + lane; k[j] = src[s + (e < n ? e : n - 1)]; }
    for (unsigned b = lane; b < bins; b += 64) h[b] = 0u;
    // (one wave: its LDS operations complete in order, no barrier needed between the fill, the adds and the read-out)
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j)
        if (j * 64 + lane < n) atomicAdd(&h[(unsigned)(k[j] >> shift) & mask], 1u);
    unsigned* __restrict__ dst = hist + q.hist_off[i] + ((size_t)tile << d);
#pragma unroll 8
    for (unsigned b = lane; b < bins; b += 64) dst[b] = h[b];
}

// one thread per (table, tile group, bin): hist[tile][bin] -> exclusive prefix over the GROUP's tiles (in place); the group's total goes to
// tot[bin] (tables of one group) or to gtot[group][bin] (long segments: seg_groupscan_kernel prefixes those).
// Neighbouring threads own neighbouring bins (coalesced), the tiles' counts are fetched 16 at a time (independent loads).
__global__ __launch_bounds__(256) void seg_colscan_kernel(SegRound q, unsigned* __restrict__ hist, unsigned* __restrict__ tot,
                                                          unsigned* __restrict__ gtot) {
    int i = 0;
    while (i + 1 < q.ntab && blockIdx.x >= q.scan_start[i + 1]) ++i;
    const int d = q.dbits[i];
    const unsigned bins = 1u << d, binblocks = (bins + 255) / 256;
    const unsigned local = blockIdx.x - q.scan_start[i];
    const unsigned grp = local / binblocks;
    const unsigned b = (local - grp * binblocks) * 256 + threadIdx.x;
    if (b >= bins) return;
    const unsigned tiles_all = q.tile_start[i + 1] - q.tile_start[i];
    const unsigned tbeg = grp * SEG_GROUP_TILES;
    const unsigned tiles = tiles_all - tbeg < (unsigned)SEG_GROUP_TILES ? tiles_all - tbeg : (unsigned)SEG_GROUP_TILES;
    unsigned* __restrict__ h = hist + q.hist_off[i] + ((size_t)tbeg << d) + b;
    unsigned run = 0;
    for (unsigned t0 = 0; t0 < tiles; t0 += 16) {
        unsigned c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (t0 + u < tiles) ? h[(size_t)(t0 + u) << d] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (t0 + u < tiles) { h[(size_t)(t0 + u) << d] = run; run += c[u]; }
    }
    if (q.ngroups[i] > 1) gtot[q.gtot_off[i] + ((size_t)grp << d) + b] = run;
    else tot[q.bin_off[i] + b] = run;
}

// long segments only — one thread per (table, bin): gtot[group][bin] -> exclusive prefix over the table's groups (in place), tot[bin] = total
__global__ __launch_bounds__(256) void seg_groupscan_kernel(SegRound q, unsigned* __restrict__ gtot, unsigned* __restrict__ tot) {
    int i = 0;
    while (i + 1 < q.ntab && blockIdx.x >= q.gscan_start[i + 1]) ++i;
    const int d = q.dbits[i];
    const unsigned bins = 1u << d;
    const unsigned b = (blockIdx.x - q.gscan_start[i]) * 256 + threadIdx.x;
    if (b >= bins) return;
    const unsigned groups = q.ngroups[i];
    unsigned* __restrict__ gp = gtot + q.gtot_off[i] + b;
    unsigned run = 0;
    for (unsigned g0 = 0; g0 < groups; g0 += 16) {
        unsigned c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (g0 + u < groups) ? gp[(size_t)(g0 + u) << d] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (g0 + u < groups) { gp[(size_t)(g0 + u) << d] = run; run += c[u]; }
    }
    tot[q.bin_off[i] + b] = run;
}

// per table: tot[bin] -> exclusive prefix over the bins (in place): the first position of every digit inside the table's segment
__global__ __launch_bounds__(1024) void seg_binscan_kernel(SegRound q, unsigned* __restrict__ tot) {
    __shared__ unsigned wsum[16];
    const int i = blockIdx.x, tid = threadIdx.x;
    const unsigned bins = 1u << q.dbits[i];
    unsigned* __restrict__ t = tot + q.bin_off[i];
    const unsigned per = (bins + 1023) / 1024;                // 1 .. 8 consecutive bins per thread
    unsigned loc[8], sum = 0;
#pragma unroll
    for (unsigned j = 0; j < 8; ++j) {
        const unsigned b = tid * per + j;
        loc[j] = (j < per && b < bins) ? t[b] : 0u;
        sum += loc[j];
    }
    unsigned inc = sum;                                        // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int w = 0; w < (tid >> 6); ++w) run += wsum[w];
#pragma unroll
    for (unsigned j = 0; j < 8; ++j) {
        const unsigned b = tid * per + j;
        if (j < per && b < bins) { t[b] = run; run += loc[j]; }
    }
}

// one tile of one table: prefetch, cursors, ordered scatter.  FIRST = the table's first round (source IN, value = position: no value loads)
template <typename KT, bool FIRST>
__device__ __forceinline__ void seg_scatter_tile(const SegRound& q, int i, unsigned tile, const KT* __restrict__ ksrc,
                                                 const unsigned* __restrict__ vsrc, KT* __restrict__ kdst, unsigned* __restrict__ vdst,
                                                 const unsigned* __restrict__ hist, const unsigned* __restrict__ tot,
                                                 const unsigned* __restrict__ gtot, int dbg, unsigned* cur, unsigned char* claim) {
    const int lane = threadIdx.x;
    const int d = q.dbits[i], shift = q.shift[i];
    const unsigned bins = 1u << d, mask = bins - 1;
    const long long seg = q.base[i];
    const long long s = seg + (long long)tile * SEG_TILE;
    long long nn = q.nnz[i] - (long long)tile * SEG_TILE; if (nn > SEG_TILE) nn = SEG_TILE;
    const int n = (int)nn;
    // ---- the whole tile into registers: SEG_CHUNKS independent loads per lane (and as many for the values after the first round);
    // indices past the tile's end are clamped to its last entry, not predicated: unconditional loads in straight-line code
    KT k[SEG_CHUNKS];
    unsigned v[FIRST ? 1 : SEG_CHUNKS];
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j) { const int e = j * 64 + lane; k[j] = ksrc[s + (e < n ? e : n - 1)]; }
    if constexpr (!FIRST) {
#pragma unroll
        for (int j = 0; j < SEG_CHUNKS; ++j) { const int e = j * 64 + lane; v[j] = vsrc[s + (e < n ? e : n - 1)]; }
    }
    // ---- where the tile's entries of digit g start:  first[g] (seg_binscan_kernel) + entries of g in the earlier tiles (seg_colscan_kernel),
    // GATHERED for the digits this lane actually holds (SEG_CHUNKS independent 4-byte reads from each array) — a tile of 2048 entries
    // touches at most 2048 of up to 8192 bins, and loading + prefix-scanning all of them per tile cost more than the scatter itself
    // (35 of 49 us, profiles/round3).  LDS keeps only the running COUNT of every digit inside this tile.
    const unsigned* __restrict__ tt = tot + q.bin_off[i];
    const unsigned* __restrict__ h = hist + q.hist_off[i] + ((size_t)tile << d);
    unsigned off[SEG_CHUNKS];
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j) { const unsigned dg = (unsigned)(k[j] >> shift) & mask; off[j] = tt[dg]; }
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j) { const unsigned dg = (unsigned)(k[j] >> shift) & mask; off[j] += h[dg]; }
    if (q.ngroups[i] > 1) {                                   // long segment: + the entries of the digit in the earlier tile groups
        const unsigned* __restrict__ gg = gtot + q.gtot_off[i] + ((size_t)(tile / SEG_GROUP_TILES) << d);
#pragma unroll
        for (int j = 0; j < SEG_CHUNKS; ++j) { const unsigned dg = (unsigned)(k[j] >> shift) & mask; off[j] += gg[dg]; }
    }
#pragma unroll 16
    for (unsigned b = lane; b < bins; b += 64) cur[b] = 0u;
    __builtin_amdgcn_wave_barrier();
    // ---- 64 entries at a time, in order
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j) {
        if (j * 64 >= n) break;
        const bool valid = j * 64 + lane < n;
        const unsigned dg = (unsigned)(k[j] >> shift) & mask;
        // match-any: the set of valid lanes holding my digit.  Bitwise over the d <= 13 digit bits it took 13 ballots per step (62 of
        // the first version's 172 us per sort); instead every lane writes its number into the digit's CLAIM byte — lanes of one digit
        // hit one address and exactly one write survives — reads it back, and the lanes are matched on that 6-bit winner number.
        unsigned long long same = __ballot(valid);
        if (!(dbg & 2)) {
            if (valid) claim[dg] = (unsigned char)lane;
            __builtin_amdgcn_wave_barrier();
            const unsigned w = valid ? (unsigned)claim[dg] : 64u;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const bool bit = (w >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                same &= bit ? bal : ~bal;
            }
        }
        const int rank = __popcll(same & below);
        const int leader = __ffsll((long long)same) - 1;                 // lowest lane of my group (valid lanes only use it)
        unsigned start = 0u;
        if (valid && rank == 0 && !(dbg & 4)) {                          // one lane per distinct digit: no two leaders share an address
            start = cur[dg];
            cur[dg] = start + (unsigned)__popcll(same);
        }
        __builtin_amdgcn_wave_barrier();
        start = off[j] + __shfl(start, leader < 0 ? 0 : leader, 64);    // (off[j] is the same for every lane of the group)
        const unsigned val = FIRST ? (unsigned)(s + j * 64 + lane) : v[FIRST ? 0 : j];
        if (valid && !(dbg & 1)) {
            const long long dst = seg + (long long)start + rank;
            kdst[dst] = k[j];
            vdst[dst] = val;
        }
        if ((dbg & 1) && start + rank + k[j] + val == 0x7fffffffu) kdst[0] = 0;   // keeps the values live
    }
}

template <typename KT>
__global__ __launch_bounds__(64) void seg_scatter_kernel(SegRound q, const KT* __restrict__ kin, const KT* __restrict__ ktmp_r,
                                                         const KT* __restrict__ kout_r, KT* __restrict__ ktmp, KT* __restrict__ kout,
                                                         const unsigned* __restrict__ vtmp_r, const unsigned* __restrict__ vout_r,
                                                         unsigned* __restrict__ vtmp, unsigned* __restrict__ vout,
                                                         const unsigned* __restrict__ hist, const unsigned* __restrict__ tot,
                                                         const unsigned* __restrict__ gtot, int dbg) {
    // dbg (env DLRM_SEG_DEBUG, timing only — WRONG results): 1 no scattered stores, 2 no match-any, 4 no LDS cursor hand-over.
    // Lanes hand cursors to each other through `cur`.  One wave: its LDS instructions execute in program order; the
    // __builtin_amdgcn_wave_barrier() calls keep the COMPILER from moving LDS accesses across the hand-over points.
    __shared__ unsigned cur[1 << SEG_MAX_DBITS];            // per digit: entries of it placed so far in this tile
    __shared__ unsigned char claim[1 << SEG_MAX_DBITS];     // per digit: the lane that claimed it in the current step (match-any label)
    const int i = seg_find(q, blockIdx.x);
    const unsigned tile = blockIdx.x - q.tile_start[i];
    const bool to_out = q.dst_out[i] != 0;
    KT* __restrict__ kdst = to_out ? kout : ktmp;
    unsigned* __restrict__ vdst = to_out ? vout : vtmp;
    if (q.first[i])
        seg_scatter_tile<KT, true>(q, i, tile, kin, nullptr, kdst, vdst, hist, tot, gtot, dbg, cur, claim);
    else
        seg_scatter_tile<KT, false>(q, i, tile, to_out ? ktmp_r : kout_r, to_out ? vtmp_r : vout_r, kdst, vdst, hist, tot, gtot, dbg, cur, claim);
}

static int seg_debug() {
    static int v = -1;
    if (v < 0) v = DLRM_DEBUG_ENV("DLRM_SEG_DEBUG", 0x7fffffff);
    return v;
}

template <typename KT>
static int seg_sort_run(const SegPlan& p, const KT* keys_in, KT* keys_tmp, KT* keys_out, unsigned* vals_tmp, unsigned* vals_out,
                        unsigned* hist, unsigned* binbase, unsigned* gtot, hipStream_t st) {
    for (int r = 0; r < p.rounds; ++r) {
        const SegRound& q = p.round[r];
        const unsigned tiles = q.tile_start[q.ntab];
        if (q.ntab == 0 || tiles == 0) continue;
        hipLaunchKernelGGL((seg_hist_kernel<KT>), dim3(tiles), dim3(64), 0, st, q, keys_in, (const KT*)keys_tmp, (const KT*)keys_out, hist);
        DLRM_LAUNCH_CHECK();
        hipLaunchKernelGGL(seg_colscan_kernel, dim3(q.scan_start[q.ntab]), dim3(256), 0, st, q, hist, binbase, gtot);
        DLRM_LAUNCH_CHECK();
        if (q.gscan_start[q.ntab] > 0) {
            hipLaunchKernelGGL(seg_groupscan_kernel, dim3(q.gscan_start[q.ntab]), dim3(256), 0, st, q, gtot, binbase);
            DLRM_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(seg_binscan_kernel, dim3((unsigned)q.ntab), dim3(1024), 0, st, q, binbase);
        DLRM_LAUNCH_CHECK();
        hipLaunchKernelGGL((seg_scatter_kernel<KT>), dim3(tiles), dim3(64), 0, st, q, keys_in, (const KT*)keys_tmp, (const KT*)keys_out,
                           keys_tmp, keys_out, (const unsigned*)vals_tmp, (const unsigned*)vals_out, vals_tmp, vals_out,
                           (const unsigned*)hist, (const unsigned*)binbase, (const unsigned*)gtot, seg_debug());
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace
