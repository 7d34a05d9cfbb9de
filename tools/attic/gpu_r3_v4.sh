#!/bin/bash
# visit 4b: claim-based match-any + batched cursor fills — exactness, then timing (debug switches: wrong results by design)
OUT=gpurun_out/v4; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -p no:cacheprovider -k "lookup_sort or emb_bwd_sgd_sorted" 2>&1 | tail -3
for dbg in 0 1 2 4 7; do
  DLRM_SEG_DEBUG=$dbg timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1 | sed "s/^/dbg=$dbg  /"
done
DLRM_SORT=rocprim timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1
timeout 120 python tools/sort_bench.py 2048 2>&1 | tail -1
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_sort -o s -- python $GRAFT_REPO_ROOT/tools/sort_bench.py 65536 > $GRAFT_REPO_ROOT/$OUT/rocprof_sort.log 2>&1 )
f=$(find $OUT/rocprof_sort -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "seg_" in n or "expand" in n: print("%-40s calls %4s avg %8.1f us" % (n.split("(anonymous namespace)::")[1][:38] if "(anonymous namespace)::" in n else n[:38], r["Calls"], float(r["AverageNs"])/1e3))
PY
find $OUT/rocprof_sort -name "*kernel_trace.csv" -size +4M -delete
