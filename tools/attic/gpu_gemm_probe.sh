#!/bin/bash
# where does the fp32 GEMM lose its 20 %?  per-layer microbench with the tuning switches of gemm3_kernel + SQ counters
OUT=gpurun_out/${1:-gemmprobe}; mkdir -p $OUT
for f in 0 1 2 4 3 7; do echo "== DLRM_GEMM_DEBUG=$f"; DLRM_GEMM_DEBUG=$f timeout 300 python tools/microbench.py gemm > $OUT/micro_dbg$f.log 2>&1; grep gemm $OUT/micro_dbg$f.log | cut -c1-230; done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$tag -o p -- python $GRAFT_REPO_ROOT/tools/microbench.py gemm_big > $GRAFT_REPO_ROOT/$OUT/$tag.log 2>&1
  echo "rc=$? $tag"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm3" not in n: continue
        key = n.split("gemm3_kernel")[1][:28] + " grid " + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items()):
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
