#!/bin/bash
OUT=gpurun_out/${1:-xcd}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "(linear and not 65536 and not 66000) or terabyte or training_matches or mlp" 2>&1 | tail -3
bash tools/gpu_ab.sh $1/ab DLRM_GEMM_DEBUG=0 DLRM_GEMM_DEBUG=32
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap"
cd /tmp
for dbg in 0 32; do
  DLRM_GEMM_DEBUG=$dbg timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/F$dbg/FETCH_SIZE -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/F$dbg.log 2>&1
  echo "rc=$? FETCH dbg=$dbg"
done
cd $GRAFT_REPO_ROOT
for dbg in 0 32; do python tools/pmc_fold.py $OUT/F$dbg 2>/dev/null | head -3; grep "gemm3_kernel<false, false" $OUT/F$dbg/pmc_FETCH_SIZE.csv; done
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete
