#!/bin/bash
# Per-kernel times of the lookup sort (tools/sort_bench.py under rocprofv3) for one or more builds of the library.
#   usage: tools/sort_prof.sh <out-dir> [<tag> ...]     tag = suffix of dlrm_amd/libdlrm_hip_<tag>.so ("head" = the product library)
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
for tag in "$@"; do
  lib=$ROOT/dlrm_amd/libdlrm_hip_$tag.so; [ "$tag" = head ] && lib=$ROOT/dlrm_amd/libdlrm_hip.so
  rm -rf /tmp/sp_$tag
  ( cd /tmp && DLRM_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$tag -o s -- python $ROOT/tools/sort_bench.py > /tmp/sp_$tag.log 2>&1 )
  f=$(find /tmp/sp_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $ROOT/$OUT/sort_kernel_stats_$tag.csv
  echo "== $tag: $(grep sort_lookups /tmp/sp_$tag.log)"
  python - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "seg_" in n or "expand" in n:
        k = n.split("::")[-1].split("(")[0].split("<")[0]
        per_sort = float(r["TotalDurationNs"]) / 55 / 1e3
        tot += per_sort
        print("   %-22s calls/sort %.0f  avg %.2f us  per sort %.2f us" % (k, int(r["Calls"]) / 55, float(r["AverageNs"]) / 1e3, per_sort))
print("   sum per sort %.1f us" % tot)
PY
done
