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
constexpr int SEG_BLOCKS = (1 << SEG_MAX_DBITS) / 256;   // 256-bin blocks of the widest digit: per-tile block sums [tile][SEG_BLOCKS] (seg_scan_kernel)
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
    size_t hist_words, bin_words, gtot_words, bsum_words;    // buffer sizes (u32 words), the maximum over rounds
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
    p->hist_words = 0; p->bin_words = 0; p->gtot_words = 0; p->bsum_words = 0;
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
            hw = (hw + 3) & ~(size_t)3; bw = (bw + 3) & ~(size_t)3; gw = (gw + 3) & ~(size_t)3;   // rows of the next table start 16-byte aligned (seg_scatter_tile)
        }
        for (int i = q.ntab; i < DLRM_MAX_TABLES_PER_LAUNCH; ++i) {
            q.tab[i] = 0; q.tile_start[i + 1] = q.tile_start[q.ntab]; q.scan_start[i + 1] = q.scan_start[q.ntab]; q.gscan_start[i + 1] = q.gscan_start[q.ntab];
            q.dbits[i] = 1; q.shift[i] = 0; q.first[i] = 0; q.dst_out[i] = 1;
            q.hist_off[i] = 0; q.bin_off[i] = 0; q.ngroups[i] = 1; q.gtot_off[i] = 0; q.base[i] = 0; q.nnz[i] = 0;
        }
        if (hw > p->hist_words) p->hist_words = hw;
        if (bw > p->bin_words) p->bin_words = bw;
        if (gw > p->gtot_words) p->gtot_words = gw;
        if ((size_t)q.tile_start[q.ntab] * SEG_BLOCKS > p->bsum_words) p->bsum_words = (size_t)q.tile_start[q.ntab] * SEG_BLOCKS;
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

// tools/probes/seg_scatter_probe.hip only: cycle stamps of a scatter tile's phases (each stamp waits for everything issued before it,
// so the stamped build serialises what the product build overlaps: it prices the phases, it is not the product's timeline)
#ifdef DLRM_SEG_STAMPS
__device__ unsigned long long seg_stamps[8192 * 8];
#define SEG_STAMP(n) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                          if (threadIdx.x == 0) seg_stamps[(size_t)blockIdx.x * 8 + (n)] = __builtin_readcyclecounter(); } while (0)
#else
#define SEG_STAMP(n) do { } while (0)
#endif

constexpr int SEG_CHUNKS = SEG_TILE / 64;     // 64-entry steps per tile: a lane keeps its SEG_CHUNKS keys of the tile in registers

// All loads of a tile are issued up front (SEG_CHUNKS independent, coalesced loads per lane): a wave that fetched one 64-entry step
// at a time paid a full memory round trip per step — 64 dependent round trips per tile, ~0.2 ms per pass on Criteo-Terabyte shapes
// (profiles/round3: the first version of this sorter was slower than rocPRIM for exactly that reason).
template <typename KT>
__global__ __launch_bounds__(64) void seg_hist_kernel(SegRound q, const KT* __restrict__ in, const KT* __restrict__ tmp,
                                                      const KT* __restrict__ out, unsigned* __restrict__ hist, unsigned* __restrict__ bsum) {
    __shared__ __attribute__((aligned(16))) unsigned h[1 << SEG_MAX_DBITS];
    __shared__ unsigned bs[SEG_BLOCKS];                       // the tile's entries per block of 256 bins (what seg_scan_kernel needs of the OTHER blocks)
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
    // (one wave: its LDS operations complete in order, no barrier needed between the fill, the adds and the read-out; rows of >= 256 bins
    // move 16 bytes per lane — the table's rows are 16-byte aligned, seg_plan)
    if (d >= 8) { for (unsigned b = lane * 4; b < bins; b += 256) *(uint4*)(h + b) = make_uint4(0u, 0u, 0u, 0u); }
    else        { for (unsigned b = lane; b < bins; b += 64) h[b] = 0u; }
    if (lane < SEG_BLOCKS) bs[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < SEG_CHUNKS; ++j)
        if (j * 64 + lane < n) {
            const unsigned dg = (unsigned)(k[j] >> shift) & mask;
            atomicAdd(&h[dg], 1u);
            atomicAdd(&bs[dg >> 8], 1u);
        }
    __builtin_amdgcn_wave_barrier();
    if (lane < SEG_BLOCKS) bsum[(size_t)blockIdx.x * SEG_BLOCKS + lane] = bs[lane];
    unsigned* __restrict__ dst = hist + q.hist_off[i] + ((size_t)tile << d);
    if (d >= 8) {
#pragma unroll 8
        for (unsigned b = lane * 4; b < bins; b += 256) *(uint4*)(dst + b) = *(const uint4*)(h + b);
    } else {
        for (unsigned b = lane; b < bins; b += 64) dst[b] = h[b];
    }
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

// Rounds without long segments (every one-lookup-per-bag batch): seg_colscan_kernel AND seg_binscan_kernel in one launch (round 6; two
// launches of ~9.5 us each for a few hundred KB of counters).  One workgroup per (table, block of 256 bins), one thread per bin:
//   hist[tile][bin] -> exclusive prefix over the table's tiles (in place, as seg_colscan_kernel), the bin's total stays in a register;
//   first position of the bin = entries of the table in EARLIER blocks (summed from the per-tile block sums seg_hist_kernel wrote:
//   at most 128 tiles x 31 blocks words) + exclusive prefix of the totals inside the block (wave scans + the four waves' sums)
// -> tot[bin], final: what seg_binscan_kernel produced.  Integer sums: any order gives the same words.
__global__ __launch_bounds__(256) void seg_scan_kernel(SegRound q, unsigned* __restrict__ hist, unsigned* __restrict__ tot,
                                                       const unsigned* __restrict__ bsum) {
    __shared__ unsigned wtot[4], wbase[4];
    int i = 0;
    while (i + 1 < q.ntab && blockIdx.x >= q.scan_start[i + 1]) ++i;
    const int d = q.dbits[i], tid = threadIdx.x;
    const unsigned bins = 1u << d;
    const unsigned bb = blockIdx.x - q.scan_start[i];         // (one tile group: scan_start counts 256-bin blocks)
    const unsigned b = bb * 256 + tid;
    const unsigned tiles = q.tile_start[i + 1] - q.tile_start[i];
    // entries of the table in the blocks before mine: thread -> (tile = tid / 32 + 8 r, block = tid % 32), rows of 32 words read whole
    unsigned before = 0;
    if ((unsigned)(tid & 31) < bb) {
        const unsigned* __restrict__ bp = bsum + (size_t)q.tile_start[i] * SEG_BLOCKS + (tid & 31);
        for (unsigned t = tid >> 5; t < tiles; t += 8) before += bp[(size_t)t * SEG_BLOCKS];
    }
    unsigned run = 0;
    if (b < bins) {
        unsigned* __restrict__ h = hist + q.hist_off[i] + b;
        for (unsigned t0 = 0; t0 < tiles; t0 += 16) {
            unsigned c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = (t0 + u < tiles) ? h[(size_t)(t0 + u) << d] : 0u;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (t0 + u < tiles) { h[(size_t)(t0 + u) << d] = run; run += c[u]; }
        }
    }
    unsigned inc = run, bsumw = before;                       // inclusive scan of the totals / plain sum of `before` over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += v;
        bsumw += __shfl_xor(bsumw, o, 64);
    }
    if ((tid & 63) == 63) { wtot[tid >> 6] = inc; wbase[tid >> 6] = bsumw; }
    __syncthreads();
    unsigned first = inc - run + wbase[0] + wbase[1] + wbase[2] + wbase[3];
    for (int w = 0; w < (tid >> 6); ++w) first += wtot[w];
    if (b < bins) tot[q.bin_off[i] + b] = first;
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
    SEG_STAMP(1);
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
    SEG_STAMP(2);
    // ---- where the tile's entries of digit g start:  first[g] (seg_binscan_kernel) + entries of g in the earlier tiles (seg_colscan_kernel)
    // [+ in the earlier tile groups of a long segment] = the INITIAL VALUE of the digit's cursor in LDS: all bins of the table are filled, by
    // coalesced 16-byte loads of the two (three) rows.  (Rounds 3-5 gathered the two words for the digits a lane holds — 2 x SEG_CHUNKS
    // divergent 4-byte loads per lane — and added them to a cursor that counted from zero; measured equal, 26.4 vs 27.0 us per launch,
    // profiles/round6/sort_kernels.md: the fill needs no per-entry registers and no add after the hand-over.)
    const unsigned* __restrict__ tt = tot + q.bin_off[i];
    const unsigned* __restrict__ h = hist + q.hist_off[i] + ((size_t)tile << d);
    const unsigned* __restrict__ gg = q.ngroups[i] > 1 ? gtot + q.gtot_off[i] + ((size_t)(tile / SEG_GROUP_TILES) << d) : nullptr;
    if (d >= 8) {                                             // whole rows of 256 bins: a lane owns 4 consecutive bins per sweep (rows are 16-byte aligned: seg_plan)
        for (unsigned b0 = 0; b0 < bins; b0 += 256 * 8) {
            uint4 a[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned b = b0 + u * 256 + lane * 4;
                if (b < bins) { a[u] = *(const uint4*)(tt + b); c[u] = *(const uint4*)(h + b); }
            }
            if (gg) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned b = b0 + u * 256 + lane * 4;
                    if (b < bins) { const uint4 g = *(const uint4*)(gg + b); a[u].x += g.x; a[u].y += g.y; a[u].z += g.z; a[u].w += g.w; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned b = b0 + u * 256 + lane * 4;
                if (b < bins) *(uint4*)(cur + b) = make_uint4(a[u].x + c[u].x, a[u].y + c[u].y, a[u].z + c[u].z, a[u].w + c[u].w);
            }
        }
    } else {
        for (unsigned b = lane; b < bins; b += 64) cur[b] = tt[b] + h[b] + (gg ? gg[b] : 0u);
    }
    __builtin_amdgcn_wave_barrier();
    SEG_STAMP(3);
    const unsigned long long below = (1ull << lane) - 1ull;
    // ---- 64 entries at a time IN ORDER — but no step waits for another (round 6).  The first version ran the steps one after the other,
    // each paying its own LDS round trips (claim byte back, cursor read, cursor write, shuffle) with one wave per SIMD and nothing to hide
    // them behind: 36-39 us per launch, now 26-27 (profiles/round6/sort_kernels.md).  A wave's LDS instructions execute in program order, so the ORDER between steps needs
    // no waiting at all: four passes over the tile's steps, each a run of independent LDS instructions issued back to back.
    // (dbg, tuning builds: 1 = no scattered stores)
    // pass A: match-any labels.  Every lane writes its number into its digit's CLAIM byte (lanes of one digit hit one address, one write
    // survives) and reads the winner back; the next step's writes are queued behind this step's reads.
    // pass B (registers only): the lanes holding my digit = the lanes that read my winner: 6 ballots over the winner's bits.
    // pass C: the WINNER of every group takes the digit's cursor and advances it by the group's size: ONE returning LDS add per step
    // (no two winners of a step share an address; the adds of consecutive steps on one digit execute in step order).
    // pass D: the group's start from its winner, then the entry goes to start + its rank in the group (lanes of the group below it):
    // stable by construction.
    // The passes run over SEG_PIPE steps at a time (the LDS queue holds 16 instructions; more steps per pass only cost registers).
    constexpr int SEG_PIPE = 8;
    static_assert(SEG_CHUNKS % SEG_PIPE == 0, "tile = whole pipeline groups");
#pragma unroll
    for (int j0 = 0; j0 < SEG_CHUNKS; j0 += SEG_PIPE) {
        if (j0 * 64 >= n) break;
        unsigned w[SEG_PIPE];                                 // the lane whose claim survived = the group's leader (any agreed member will do)
        unsigned pk[SEG_PIPE];                                // my rank in the group | group size << 8
        unsigned st[SEG_PIPE];
#pragma unroll
        for (int u = 0; u < SEG_PIPE; ++u) {                  // ---- pass A
            const int j = j0 + u;
            const bool valid = j * 64 + lane < n;
            const unsigned dg = (unsigned)(k[j] >> shift) & mask;
            if (valid) claim[dg] = (unsigned char)lane;
            __builtin_amdgcn_wave_barrier();
            w[u] = valid ? (unsigned)claim[dg] : 64u;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int u = 0; u < SEG_PIPE; ++u) {                  // ---- pass B: same = lanes whose winner equals mine, as two 32-bit halves
            const unsigned long long bv = __ballot(w[u] < 64u);
            unsigned slo = (unsigned)bv, shi = (unsigned)(bv >> 32);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int bm = (int)(w[u] << (31 - b)) >> 31;            // bit b of my winner, spread over the word (0 / -1)
                const unsigned long long bal = __ballot(bm != 0);
                slo &= ~((unsigned)bal ^ (unsigned)bm);                  // lanes whose bit b equals mine
                shi &= ~((unsigned)(bal >> 32) ^ (unsigned)bm);
            }
            const unsigned rank = __builtin_amdgcn_mbcnt_hi(shi, __builtin_amdgcn_mbcnt_lo(slo, 0u));   // lanes of my group below me
            pk[u] = rank | ((unsigned)(__popc(slo) + __popc(shi)) << 8);
        }
#pragma unroll
        for (int u = 0; u < SEG_PIPE; ++u) {                  // ---- pass C
            const unsigned dg = (unsigned)(k[j0 + u] >> shift) & mask;
            st[u] = 0u;
            if (w[u] == (unsigned)lane) st[u] = atomicAdd(&cur[dg], pk[u] >> 8);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < SEG_PIPE; ++u)                    // ---- pass D: starts from the winners ...
            st[u] = __shfl(st[u], (int)(w[u] & 63u), 64);
#pragma unroll
        for (int u = 0; u < SEG_PIPE; ++u) {                  // ... and the stores
            const int j = j0 + u;
            const unsigned val = FIRST ? (unsigned)(s + j * 64 + lane) : v[FIRST ? 0 : j];
            if (w[u] < 64u && !(dbg & 1)) {
                const long long dst = seg + (long long)(st[u] + (pk[u] & 63u));
                kdst[dst] = k[j];
                vdst[dst] = val;
            }
        }
#ifdef DLRM_SEG_STAMPS
        if (j0 == SEG_CHUNKS - SEG_PIPE) { if (threadIdx.x == 0) seg_stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_readcyclecounter(); }
#endif
    }
    SEG_STAMP(5);
}

template <typename KT>
__global__ __launch_bounds__(64) void seg_scatter_kernel(SegRound q, const KT* __restrict__ kin, const KT* __restrict__ ktmp_r,
                                                         const KT* __restrict__ kout_r, KT* __restrict__ ktmp, KT* __restrict__ kout,
                                                         const unsigned* __restrict__ vtmp_r, const unsigned* __restrict__ vout_r,
                                                         unsigned* __restrict__ vtmp, unsigned* __restrict__ vout,
                                                         const unsigned* __restrict__ hist, const unsigned* __restrict__ tot,
                                                         const unsigned* __restrict__ gtot, int dbg) {
    // dbg (env DLRM_SEG_DEBUG, tuning builds, timing only — WRONG results): 1 no scattered stores.
    // Lanes hand cursors to each other through `cur`.  One wave: its LDS instructions execute in program order; the
    // __builtin_amdgcn_wave_barrier() calls keep the COMPILER from moving LDS accesses across the hand-over points.
    __shared__ __attribute__((aligned(16))) unsigned cur[1 << SEG_MAX_DBITS];   // per digit: where its next entry of this tile goes (position inside the table's segment)
    __shared__ unsigned char claim[1 << SEG_MAX_DBITS];     // per digit: the lane that claimed it in the current step (match-any label)
    SEG_STAMP(0);
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
                        unsigned* hist, unsigned* binbase, unsigned* gtot, unsigned* bsum, hipStream_t st) {
    for (int r = 0; r < p.rounds; ++r) {
        const SegRound& q = p.round[r];
        const unsigned tiles = q.tile_start[q.ntab];
        if (q.ntab == 0 || tiles == 0) continue;
        hipLaunchKernelGGL((seg_hist_kernel<KT>), dim3(tiles), dim3(64), 0, st, q, keys_in, (const KT*)keys_tmp, (const KT*)keys_out, hist, bsum);
        DLRM_LAUNCH_CHECK();
        if (q.gscan_start[q.ntab] == 0) {                     // no long segment in this round: tile prefixes and digit starts from ONE launch
            hipLaunchKernelGGL(seg_scan_kernel, dim3(q.scan_start[q.ntab]), dim3(256), 0, st, q, hist, binbase, (const unsigned*)bsum);
            DLRM_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(seg_colscan_kernel, dim3(q.scan_start[q.ntab]), dim3(256), 0, st, q, hist, binbase, gtot);
            DLRM_LAUNCH_CHECK();
            hipLaunchKernelGGL(seg_groupscan_kernel, dim3(q.gscan_start[q.ntab]), dim3(256), 0, st, q, gtot, binbase);
            DLRM_LAUNCH_CHECK();
            hipLaunchKernelGGL(seg_binscan_kernel, dim3((unsigned)q.ntab), dim3(1024), 0, st, q, binbase);
            DLRM_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL((seg_scatter_kernel<KT>), dim3(tiles), dim3(64), 0, st, q, keys_in, (const KT*)keys_tmp, (const KT*)keys_out,
                           keys_tmp, keys_out, (const unsigned*)vals_tmp, (const unsigned*)vals_out, vals_tmp, vals_out,
                           (const unsigned*)hist, (const unsigned*)binbase, (const unsigned*)gtot, seg_debug());
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace
