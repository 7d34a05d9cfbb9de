#!/bin/bash
# Round-3 GPU visit: smoke, GPU parity tests, bench (with parity_check), rocprof kernel stats at HEAD, the two PMC passes.
# Usage: gpurun --timeout 1700 -- 'bash tools/gpu_round3.sh [tag] [pytest -k expr]'
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== device"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -8
  python -c "import torch,os;print('torch',torch.__version__,'gpus',torch.cuda.device_count(),'cpus',os.cpu_count())"
  python -c "from dlrm_amd import ops; print(ops.device_info(0))"
} > $OUT/device.log 2>&1
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=8 ${2:+-k "$2"} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
echo "== bench";  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; grep -v amdgpu.ids $OUT/bench.err | tail -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], d.get("parity_check")))
    for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
    print("alt", d.get("alt_mlp_arith")); print("cpu", d.get("cpu_baseline"))
except Exception as e: print("no bench json", e)
PY
echo "== rocprof"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
f=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f"
find $OUT/rocprof -name "*kernel_trace.csv" -size +20M -delete
if [ "${3:-pmc}" = "pmc" ]; then
echo "== pmc"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete
fi
export DLRM_BENCH_WATCHDOG=120
extra() { name=$1; shift; timeout 400 python bench.py "$@" --no-alt-arith > $OUT/$name.json 2> $OUT/$name.err; echo "== $name rc=$?"; grep -v amdgpu.ids $OUT/$name.err | tail -3 | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); print("  value %.0f  ms %.4f  loss %.5f dtype %s" % (d["value"], d["ms_per_step"], d["final_loss"], d["dtype"]))
except Exception as e: print("  no json", e)
PY
}
extra bench_overlap --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --overlap
extra tb_graph --steps 20 --warmup 5 --graph --no-cpu-baseline --no-parity-check
extra kaggle_eager --workload criteo_kaggle --steps 200 --warmup 10 --no-kernel-timers --no-cpu-baseline
extra kaggle_graph --workload criteo_kaggle --steps 200 --warmup 10 --graph --no-cpu-baseline
extra tb_rwsadagrad --steps 20 --warmup 5 --optimizer rwsadagrad --lr 0.0001 --no-cpu-baseline --no-parity-check
extra tb_bf16 --steps 20 --warmup 5 --mlp-arith bf16 --no-cpu-baseline --no-parity-check
extra tb_bf16x6 --steps 20 --warmup 5 --mlp-arith bf16x6 --no-cpu-baseline
extra mlperf_v2_multihot_dcn --workload mlperf_v2_multihot --steps 10 --warmup 3
extra mlperf_v2_multihot_dot --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3
du -sh $OUT
