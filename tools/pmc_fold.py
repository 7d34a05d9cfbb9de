#!/usr/bin/env python3
"""Fold rocprofv3 `p_counter_collection.csv` files of the FETCH_SIZE / WRITE_SIZE passes (tools/gpu_round3.sh) into the small
per-kernel averages pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv that tools/pmc_to_json.py reads.  usage: pmc_fold.py <dir>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file for", c)
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocclr" in n:
            continue
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0][:90]
        if "rocprim" in n:
            n = "rocprim::" + ("onesweep_iteration" if "onesweep_iteration" in r["Kernel_Name"] else "histogram")
        agg[(n, r["Grid_Size"], r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(os.path.join(out, f"pmc_{c}.csv"), "w") as f:
        f.write("kernel,grid_threads,counter,calls,avg_value,avg_us\n")
        for (n, g, cn), v in sorted(agg.items()):
            f.write('"%s",%s,%s,%d,%.6g,%.2f\n' % (n, g, cn, len(v), sum(x for x, _ in v) / len(v), sum(d for _, d in v) / len(v) / 1e3))
    print(c, "kernels:", len(agg))
