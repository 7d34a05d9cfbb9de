#!/bin/bash
# A/B of the fragment-read schedules of the fp32 GEMM (tuning build: DLRM_GEMM_SCHED = fwd/dgrad/wgrad digits) inside one visit
OUT=gpurun_out/${1:-sched}; mkdir -p $OUT
export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for r in 1 2; do
  for s in ${SCHEDS:-001 222 111}; do
    DLRM_GEMM_SCHED=$s python tools/gemm_forms_bench.py $( [ $r = 1 ] && echo --check ) > $OUT/sched_${s}_r$r.log 2>&1
    echo "== SCHED=$s round $r"; grep -E "TOTAL|maxerr [1-9]\.[0-9]+e-0[0-3]" $OUT/sched_${s}_r$r.log
  done
done
python - $OUT <<'PY'
import sys, glob, re, collections
out = sys.argv[1]
t = collections.defaultdict(dict); sums = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/sched_*_r*.log")):
    s = re.search(r"sched_(\d+)_r(\d)", f).groups()
    for l in open(f):
        m = re.match(r"(\S+)\s+(fwd|dgrad|wgrad)\s+([\d.]+) us\s+([\d.]+) TF\s+sum (\w+)", l)
        if m:
            t[(m.group(1), m.group(2))].setdefault(s[0], []).append(float(m.group(3)))
            sums[(m.group(1), m.group(2))][s[0]] = m.group(5)
scheds = sorted({k for v in t.values() for k in v})
print("| layer | form | " + " | ".join("SCHED=%s µs (min of rounds)" % s for s in scheds) + " | checksums equal |")
print("|---|---|" + "---:|" * len(scheds) + "---|")
for k in t:
    print("| %s | %s | " % k + " | ".join("%.1f" % min(t[k][s]) for s in scheds) + " | %s |" % (len(set(sums[k].values())) == 1))
PY
