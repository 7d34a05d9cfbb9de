#!/bin/bash
# HBM traffic of the hot-path kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), kernel-trace only.  Output: gpurun_out/<tag>/pmc_*.csv
TAG=${1:-pmc01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-arith > $OUT/$c.log 2>&1
  echo "rc=$? $c"
  python - "$OUT/$c/p_counter_collection.csv" "$OUT/pmc_$c.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n: continue
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:90]
    if "rocprim" in n: n = "rocprim::" + ("onesweep_iteration" if "onesweep_iteration" in r["Kernel_Name"] else "histogram")
    agg[(n, r["Grid_Size"], r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_threads,counter,calls,avg_value,avg_us\n")
    for (n, g, c), v in sorted(agg.items()):
        f.write('"%s",%s,%s,%d,%.6g,%.2f\n' % (n, g, c, len(v), sum(x for x, _ in v) / len(v), sum(d for _, d in v) / len(v) / 1e3))
PY
  rm -rf $OUT/$c
done
ls -la $OUT
