// seg_scatter_probe.hip — where a launch of seg_scatter_kernel (csrc/seg_sort.h) spends its time.  Round 6: pipelining the cursor steps took the
// launch from 38 to 27 us, replacing the per-key gathers by a coalesced cursor fill changed nothing, tiles of 1024 / 2048 / 4096 entries
// differ by a few us: no single suspect.  This probe runs the library's own sorter (the header is included as is, with DLRM_SEG_STAMPS) on
// the Criteo-Terabyte one-hot batch and prints, per round, the cycle stamps of the tiles' phases:
//   0 kernel entry  1 table found, scalars loaded  2 keys (and values) arrived  3 cursors filled  4 last pass issued  5 stores drained
// Every stamp waits for all earlier memory operations: the phases are priced one after the other (the product build overlaps 2 and 3).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDLRM_SEG_STAMPS -Idlrm_amd/csrc -Iinclude tools/probes/seg_scatter_probe.hip -o tools/probes/seg_scatter_probe
#include "seg_sort.h"
#include <algorithm>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long ROWS[26] = {39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
                                4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36};
    const int T = 26; const long long B = 65536;
    long long nnz[32], rows[32];
    for (int t = 0; t < T; ++t) { nnz[t] = B; rows[t] = ROWS[t]; }
    SegPlan plan;
    if (!seg_plan(T, nnz, rows, &plan, false)) { fprintf(stderr, "no plan\n"); return 1; }
    const size_t L = (size_t)T * B;
    std::vector<unsigned> keys(L);
    std::mt19937_64 rng(1);
    for (int t = 0; t < T; ++t) for (long long b = 0; b < B; ++b) keys[t * B + b] = (unsigned)(rng() % (unsigned long long)ROWS[t]);
    unsigned *kin, *ktmp, *kout, *vtmp, *vout, *hist, *bin, *gtot, *bsum;
    CK(hipMalloc(&kin, L * 4)); CK(hipMalloc(&ktmp, L * 4)); CK(hipMalloc(&kout, L * 4)); CK(hipMalloc(&vtmp, L * 4)); CK(hipMalloc(&vout, L * 4));
    CK(hipMalloc(&hist, plan.hist_words * 4 + 256)); CK(hipMalloc(&bin, plan.bin_words * 4 + 256)); CK(hipMalloc(&gtot, plan.gtot_words * 4 + 256));
    CK(hipMalloc(&bsum, plan.bsum_words * 4 + 256));
    CK(hipMemcpy(kin, keys.data(), L * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int it = 0; it < 5; ++it) if (seg_sort_run<unsigned>(plan, kin, ktmp, kout, vtmp, vout, hist, bin, gtot, bsum, st)) return 1;
    CK(hipStreamSynchronize(st));
    // one round at a time, stamps read back after each scatter launch
    for (int r = 0; r < plan.rounds; ++r) {
        SegPlan one = plan; one.rounds = r + 1;
        // (rounds 0..r again: the stamps left behind are those of round r's scatter, the last launch)
        if (seg_sort_run<unsigned>(one, kin, ktmp, kout, vtmp, vout, hist, bin, gtot, bsum, st)) return 1;
        CK(hipStreamSynchronize(st));
        const unsigned tiles = plan.round[r].tile_start[plan.round[r].ntab];
        std::vector<unsigned long long> s((size_t)tiles * 8);
        CK(hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(seg_stamps), s.size() * 8));
        printf("round %d: %u tiles (ticks = shader clocks, ~2.4 GHz; the counters of different XCDs are not synchronised: per-tile differences only)\n", r, tiles);
        const char* nm[5] = {"find table + scalars", "keys (+ values) arrive", "cursor fill", "passes (32 steps)", "store drain"};
        for (int p = 0; p < 5; ++p) {
            std::vector<double> d(tiles);
            for (unsigned w = 0; w < tiles; ++w) d[w] = (double)(s[w * 8 + p + 1] - s[w * 8 + p]);
            std::sort(d.begin(), d.end());
            printf("   %-24s median %8.0f  p10 %8.0f  p90 %8.0f ticks\n", nm[p], d[tiles / 2], d[tiles / 10], d[tiles * 9 / 10]);
        }
    }
    // verify: sorted per table, stable
    std::vector<unsigned> ko(L), vo(L);
    CK(hipMemcpy(ko.data(), kout, L * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(vo.data(), vout, L * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int t = 0; t < T; ++t)
        for (long long b = 1; b < B; ++b) {
            const size_t j = t * B + b;
            if (ko[j - 1] > ko[j] || (ko[j - 1] == ko[j] && vo[j - 1] >= vo[j]) || keys[vo[j]] != ko[j]) ++bad;
        }
    printf("check: %lld order violations\n", bad);
    return bad != 0;
}
