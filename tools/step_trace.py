#!/usr/bin/env python3
"""step_trace.py <kernel_trace.csv> [step] — list the kernels of one training step (default: the last) in launch order with their durations
and the idle gap in front of each (rocprofv3 --kernel-trace csv).  Tuning aid."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
                                   int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))))
rows.sort()
# a step starts at the embedding-forward kernel or the first small-K forward GEMM: use the bce kernel as the separator
idx = [i for i, r in enumerate(rows) if r[2].startswith("bce_kernel") or "bce_kernel" in r[2]]
if len(idx) < 3:
    sys.exit("fewer than 3 steps in the trace")
# step = from just after the previous step's last kernel ... find the sgd_dense of consecutive steps
ends = [i for i, r in enumerate(rows) if "sgd_dense" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 1     # which step (index into the dense-SGD launches)
a, b = ends[k - 1] + 1, ends[k] + 1
step = rows[a:b]
t0 = step[0][0]
busy = 0
prev_end = rows[a - 1][1]
print("%9s %9s %8s  %-14s %s" % ("start us", "dur us", "gap us", "grid", "kernel"))
for s, e, n, g in step:
    print("%9.1f %9.1f %8.1f  %-14s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, n[:90]))
    busy += e - s
    prev_end = max(prev_end, e)
print("step span %.1f us, kernel time %.1f us, idle %.1f us, %d kernels" % ((step[-1][1] - rows[a - 1][1]) / 1e3, busy / 1e3,
                                                                            (step[-1][1] - rows[a - 1][1] - busy) / 1e3, len(step)))
