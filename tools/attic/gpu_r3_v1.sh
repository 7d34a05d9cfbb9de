#!/bin/bash
# Round-3 build (round 3 of the build = profiles/r04), visit 1: full GPU suite (new: 4 M-row reference fixture, mlperf_v2 parity,
# launcher on oracle/_ref), A/B of the vector-fragment GEMMs, default bench with the reference-run baseline legs.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r3_v1.sh'
OUT=gpurun_out/v1
mkdir -p $OUT
export TMPDIR=/tmp
{
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
  python -c "import torch,os,psutil;print('torch',torch.__version__,'gpus',torch.cuda.device_count(),'cpus',os.cpu_count(),'ram GB',psutil.virtual_memory().total/1e9)"
} > $OUT/device.log 2>&1
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== A/B gemm fragments (steps 30, no baselines)"
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "frag1:" "frag0:DLRM_GEMM_FRAG=0" "frag1_wtm2:DLRM_WGRAD_TM=2" "frag1b:" "frag0b:DLRM_GEMM_FRAG=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py $AB > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f  fwd %.3f dgrad %.3f wgrad %.3f  emb_bwd %.3f" % (d["ms_per_step"], k["linear_fwd"]["ms_per_step"], k["linear_bwd_data"]["ms_per_step"], k["linear_bwd_weight"]["ms_per_step"], k["emb_bwd_sgd"]["ms_per_step"]))
except Exception as e: print("$tag failed", e)
PY
done
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
echo "== bench (default)";  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; grep -v amdgpu.ids $OUT/bench.err | tail -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
    for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
    print("alt", d.get("alt_mlp_arith")); print("cpu", json.dumps(d.get("cpu_baseline"))[:600]); print("stock", json.dumps(d.get("stock_gpu_baseline"))[:600])
except Exception as e: print("no bench json", e)
PY
echo "== bench mlperf_v2 dot (parity f32 + bf16)"; timeout 600 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_v2_dot.json 2> $OUT/bench_v2_dot.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_v2_dot.json")); print("v2 dot ms %.3f" % d["ms_per_step"]); print(json.dumps(d.get("parity_check"))[:1500])
except Exception as e: print("no v2 json", e)
PY
