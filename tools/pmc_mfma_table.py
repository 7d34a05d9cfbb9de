#!/usr/bin/env python3
"""Per-kernel MFMA-busy / LDS-conflict table from the rocprofv3 PMC passes of tools/gpu_visit.sh's `pmcm` stage
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE; separate passes).   usage: pmc_mfma_table.py <dir>"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm3_kernel" in n:
            n = n[n.index("gemm3_kernel"):].split("(")[0]
        elif "rocprim" in n:
            n = "rocprim sort"
        else:
            n = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        agg[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (n, g), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rows.append((n, g, len(next(iter(c.values()))), m))
rows.sort(key=lambda r: -r[3].get("GRBM_GUI_ACTIVE", 0) * r[2])
print("| kernel | grid threads | launches | GRBM_GUI_ACTIVE | MFMA busy cycles | MFMA busy / (GUI_ACTIVE x 1024 SIMDs) | LDS bank conflict / LDS active |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, g, k, m in rows[:32]:
    gui, mf = m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    la, lc = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)
    print("| %s | %s | %d | %.0f | %.3g | %s | %s |" % (n, g, k, gui, mf, ("%.3f" % (mf / (gui * 1024))) if gui else "-", ("%.3f" % (lc / la)) if la else "-"))
