#!/bin/bash
OUT=gpurun_out/r4v8
mkdir -p $OUT
export TMPDIR=/tmp
python tools/sort_bench.py v2 2>&1 | grep sort_lookups
DLRM_SORT=rocprim python tools/sort_bench.py v2 2>&1 | grep sort_lookups
cd /tmp
for m in own rocprim; do
  DLRM_SORT=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$m -o s -- python $GRAFT_REPO_ROOT/tools/sort_bench.py v2 > /dev/null 2>&1
  st=$(find $GRAFT_REPO_ROOT/$OUT/prof_$m -name "*kernel_stats.csv" | head -1); echo "== $m"; [ -n "$st" ] && cut -d, -f1-5 "$st" | cut -c1-150 | head -14
  find $GRAFT_REPO_ROOT/$OUT/prof_$m -name "*kernel_trace.csv" -delete
done
