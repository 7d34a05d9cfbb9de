#!/bin/bash
# MFMA-pipe and LDS counters of the step's kernels (separate passes, kernel-trace only): SQ_VALU_MFMA_BUSY_CYCLES vs GRBM_GUI_ACTIVE,
# SQ_LDS_BANK_CONFLICT vs SQ_LDS_IDX_ACTIVE
OUT=gpurun_out/${1:-pmcm}; mkdir -p $OUT; export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap --no-alt-fuse --no-kernel-timers"
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/$tag.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os, sys
out = os.environ.get("OUTDIR", "gpurun_out/pmcm")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm3_kernel" in n: n = n[n.index("gemm3_kernel"):].split("(")[0]
        elif "rocprim" in n: n = "rocprim sort"
        else: n = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        agg[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (n, g), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rows.append((n, g, len(next(iter(c.values()))), m))
rows.sort(key=lambda r: -r[3].get("GRBM_GUI_ACTIVE", 0) * r[2])
print("| kernel | grid threads | launches | GRBM_GUI_ACTIVE | MFMA busy cycles | MFMA busy / (GUI_ACTIVE x 1024 SIMDs) | LDS bank conflict / LDS active |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, g, k, m in rows[:24]:
    gui, mf = m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    la, lc = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0)
    print("| %s | %s | %d | %.0f | %.3g | %s | %s |" % (n, g, k, gui, mf, ("%.3f" % (mf / (gui * 1024))) if gui else "-", ("%.3f" % (lc / la)) if la else "-"))
PY
find $OUT -name "*.csv" -size +6M -delete
