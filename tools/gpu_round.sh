#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench, rocprof kernel stats.  Everything is logged under
# gpurun_out/ (merged back by gpurun).  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== device"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -8
  python -c "import torch,os;print('torch',torch.__version__,'gpus',torch.cuda.device_count(),'cpus',os.cpu_count())"
  python -c "from dlrm_amd import ops; print(ops.device_info(0))"
} > $OUT/device.log 2>&1
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
echo "== bench";  timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-arith > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/rocprof -name "*kernel_stats*" | head -3
f=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep the merged-back payload small: the per-dispatch trace can be large
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
