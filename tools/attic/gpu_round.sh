#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench, rocprof kernel stats, extra bench modes.  Everything is logged under
# gpurun_out/ (merged back by gpurun).  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== device"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -8
  python -c "import torch,os;print('torch',torch.__version__,'gpus',torch.cuda.device_count(),'cpus',os.cpu_count())"
  python -c "from dlrm_amd import ops; print(ops.device_info(0))"
} > $OUT/device.log 2>&1
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== bench";  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json; grep -v amdgpu.ids $OUT/bench.err | tail -5
echo "== rocprof"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-arith > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
f=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f"
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
export DLRM_BENCH_WATCHDOG=60
extra() { name=$1; shift; timeout 120 python bench.py "$@" --no-cpu-baseline --no-alt-arith > $OUT/$name.json 2> $OUT/$name.err; echo "== $name rc=$?"; grep -v amdgpu.ids $OUT/$name.err | tail -4; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); print("  value %.0f  ms %.4f  loss %.5f" % (d["value"], d["ms_per_step"], d["final_loss"]), "| alt graph:", d.get("alt_hip_graph"))
except Exception as e: print("  no json", e)
PY
}
extra tb_graph --steps 20 --warmup 5 --graph
extra kaggle_eager --workload criteo_kaggle --steps 200 --warmup 10 --no-kernel-timers
extra kaggle_graph --workload criteo_kaggle --steps 200 --warmup 10 --graph
extra tb_rwsadagrad --steps 20 --warmup 5 --optimizer rwsadagrad --lr 0.0001
du -sh $OUT
