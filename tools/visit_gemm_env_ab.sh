#!/bin/bash
# A/B of one tuning-build environment knob of the fp32 GEMM inside one visit:  tools/visit_gemm_env_ab.sh <tag> <ENVNAME> <value> [<value> ...]
OUT=gpurun_out/${1:?tag}; VAR=${2:?env name}; shift 2; mkdir -p $OUT
export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for r in 1 2; do
  for s in "$@"; do
    env $VAR=$s python tools/gemm_forms_bench.py $( [ $r = 1 ] && echo --check ) > $OUT/${VAR}_${s}_r$r.log 2>&1
    echo "== $VAR=$s round $r"; grep -E "TOTAL|maxerr [1-9]\.[0-9]+e-0[0-3]|rror" $OUT/${VAR}_${s}_r$r.log
  done
done
python - $OUT $VAR <<'PY'
import sys, glob, re, collections
out, var = sys.argv[1:3]
t = collections.defaultdict(dict); sums = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/%s_*_r*.log" % var)):
    s = re.search(r"%s_(\w+)_r(\d)" % var, f).groups()
    for l in open(f):
        m = re.match(r"(\S+)\s+(fwd|dgrad|wgrad)\s+([\d.]+) us\s+([\d.]+) TF\s+sum (\w+)", l)
        if m:
            t[(m.group(1), m.group(2))].setdefault(s[0], []).append(float(m.group(3)))
            sums[(m.group(1), m.group(2))][s[0]] = m.group(5)
vals = sorted({k for v in t.values() for k in v})
print("| layer | form | " + " | ".join("%s=%s µs (min of rounds)" % (var, s) for s in vals) + " | checksums equal |")
print("|---|---|" + "---:|" * len(vals) + "---|")
tot = collections.defaultdict(float)
for k in t:
    print("| %s | %s | " % k + " | ".join("%.1f" % min(t[k][s]) for s in vals) + " | %s |" % (len(set(sums[k].values())) == 1))
    for s in vals: tot[s] += min(t[k][s])
print("| sum | | " + " | ".join("%.1f" % tot[s] for s in vals) + " | |")
PY
