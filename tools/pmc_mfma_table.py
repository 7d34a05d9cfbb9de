#!/usr/bin/env python3
"""Per-kernel MFMA-busy / LDS-conflict table from the rocprofv3 PMC passes of tools/gpu_visit.sh's `pmcm` stage
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE; separate passes).   usage: pmc_mfma_table.py <dir>

GRBM_GUI_ACTIVE: rocprofv3 of ROCm 7.2 reports the SUM over the 8 XCDs (round 3's files held one XCD's count).  The script does not assume
either: it compares the counter with the dispatch's own duration (End - Start timestamps, ns) x the shader clock and divides by the number of
XCDs that best explains it (1 or 8), printed in the `xcds` column.  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)."""
import collections
import csv
import glob
import sys

CLOCK_GHZ = 2.3
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/p_counter_collection.csv", recursive=True) + glob.glob(out + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        if "gemm3_kernel" in n:
            n = n[n.index("gemm3_kernel"):].split("(")[0]
        elif "gemm_bf16_phased_kernel" in n:
            n = n[n.index("gemm_bf16_phased_kernel"):].split("(")[0]
        elif "rocprim" in n:
            n = "rocprim sort"
        elif "at::native" in n:
            n = "torch (input generation / glue)"
        else:
            n = n.split("(")[0][:60]
        key = (n, r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
seen = set()
for key, c in agg.items():
    if key in seen:
        continue
    seen.add(key)
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rows.append((key[0], key[1], len(next(iter(c.values()))), m, sum(dur[key]) / len(dur[key])))
rows.sort(key=lambda r: -r[4] * r[2])
print("| kernel | grid threads | launches | avg us (under the profiler) | GRBM_GUI_ACTIVE | xcds | MFMA busy cycles | MFMA busy / (cycles x 1024 SIMDs) | LDS bank conflict / LDS active |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
for n, g, k, m, d_ns in rows[:32]:
    gui, mf = m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    la, lc = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)
    est = d_ns * CLOCK_GHZ
    xcds = 8 if (gui and est and gui / est > 2.8) else 1
    cyc = gui / xcds
    print("| %s | %s | %d | %.1f | %.0f | %d | %.3g | %s | %s |" % (n, g, k, d_ns / 1e3, gui, xcds, mf, ("%.3f" % (mf / (cyc * 1024))) if cyc else "-",
                                                             ("%.3f" % (lc / la)) if la else "-"))
