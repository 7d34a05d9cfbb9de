#!/bin/bash
# PMC counters of the big GEMMs (separate passes, kernel-trace only)
AR=${1:-bf16x6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_$AR
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/tools/microbench.py gemm_big --arith $AR > $OUT/$tag.log 2>&1
  echo "rc=$? $tag"
done
find $OUT -name "*.csv" | head; find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT
