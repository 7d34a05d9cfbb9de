#!/bin/bash
# several tuning-build configurations of the fp32 GEMM interleaved inside one visit:  CFGS="name:ENV=v ENV2=w;name2:..." tools/visit_gemm_cfgs.sh <tag>
OUT=gpurun_out/${1:?tag}; mkdir -p $OUT
export DLRM_HIP_LIB=${DLRM_HIP_LIB:-$PWD/dlrm_amd/libdlrm_hip_tuning.so}
IFS=';' read -ra cfgs <<< "$CFGS"
for r in 1 2; do
  for c in "${cfgs[@]}"; do
    t=${c%%:*}; e=${c#*:}
    env $e python tools/gemm_forms_bench.py > $OUT/${t}_r$r.log 2>&1
    echo "== $t round $r ($e)"; grep -E "TOTAL|rror" $OUT/${t}_r$r.log
  done
done
python - $OUT <<'PY'
import sys, glob, re, collections, os
out = sys.argv[1]
t = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/*_r[0-9].log")):
    name = os.path.basename(f).rsplit("_r", 1)[0]
    for l in open(f):
        m = re.match(r"(\S+)\s+(fwd|dgrad|wgrad)\s+([\d.]+) us", l)
        if m: t[(m.group(1), m.group(2))].setdefault(name, []).append(float(m.group(3)))
names = sorted({k for v in t.values() for k in v})
print("| layer | form | " + " | ".join(names) + " |")
print("|---|---|" + "---:|" * len(names))
tot = collections.defaultdict(float)
for k in t:
    print("| %s | %s | " % k + " | ".join("%.1f" % min(t[k][s]) for s in names) + " |")
    for s in names: tot[s] += min(t[k][s])
print("| sum | | " + " | ".join("%.1f" % tot[s] for s in names) + " |")
PY
