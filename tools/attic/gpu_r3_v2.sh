#!/bin/bash
# Round-3 build, visit 2: the library's own segmented sort (seg_sort.h) — exactness test, update tests, graph replay with the
# sorted update, A/B against rocPRIM; sharded mlperf_v2 bench control flow (N = 2 on one GPU over gloo); bf16 conversion change.
# Usage: gpurun --timeout 1200 -- 'bash tools/gpu_r3_v2.sh'
OUT=gpurun_out/v2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== sort + update + graph tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider \
  -k "lookup_sort or emb_bwd or adagrad or graphed or out_of_range or training_matches_reference_golden or rwsadagrad or bf16 or full_batch_properties or (mlperf_v2 and dot)" \
  > $OUT/pytest_sort.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_sort.log
echo "== graph probe, sorted update kept inside the captured step"
timeout 200 python tools/probes/graph_sorted_probe.py sorted > $OUT/graph_sorted_probe.log 2>&1; echo "probe rc=$?"; tail -4 $OUT/graph_sorted_probe.log
echo "== sharded multihot bench control flow + sharded model test"
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout=500 -p no:cacheprovider -k "sharded" > $OUT/pytest_sharded.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_sharded.log
echo "== A/B sort (steps 30)"
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "own:" "rocprim:DLRM_SORT=rocprim" "own_b:" "rocprim_b:DLRM_SORT=rocprim" "own_adagrad:" "rocprim_adagrad:DLRM_SORT=rocprim"; do
  tag=${cfg%%:*}; envs=${cfg#*:}; extra=""
  case $tag in *adagrad) extra="--optimizer rwsadagrad";; esac
  env $envs timeout 300 python bench.py $AB $extra > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    e = k.get("emb_bwd_sgd") or k.get("emb_bwd_adagrad")
    print("$tag ms %.3f  emb_bwd %.3f (frac %.3f)  fwd %.3f dgrad %.3f wgrad %.3f" % (d["ms_per_step"], e["ms_per_step"], e["frac"], k["linear_fwd"]["ms_per_step"], k["linear_bwd_data"]["ms_per_step"], k["linear_bwd_weight"]["ms_per_step"]))
except Exception as e: print("$tag failed", e)
PY
done
echo "== graph replay of the whole step at TB shapes (sorted update inside)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap --graph > $OUT/bench_graph.json 2> $OUT/bench_graph.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_graph.json")); print("graph ms %.3f update=%s" % (d["ms_per_step"], d["config"]["embedding_update"]))
except Exception as e: print("graph bench failed", e); print(open("$OUT/bench_graph.err").read()[-1500:])
PY
echo "== kaggle shapes, graph vs eager"
for g in "" "--graph"; do
  timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap $g > $OUT/kaggle$g.json 2> $OUT/kaggle$g.err
  python -c "
import json
try:
    d=json.load(open('$OUT/kaggle$g.json')); print('kaggle $g ms %.3f update=%s' % (d['ms_per_step'], d['config']['embedding_update']))
except Exception as e: print('kaggle $g failed', e)"
done
echo "== mlperf_v2 dot bf16 (v_cvt_pk_bf16_f32)"
timeout 600 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 --no-parity-check > $OUT/bench_v2_dot.json 2> $OUT/bench_v2_dot.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_v2_dot.json")); k=d["kernels"]; print("v2 dot ms %.3f" % d["ms_per_step"], {n: round(v["ms_per_step"],3) for n,v in k.items()})
except Exception as e: print("no v2 json", e)
PY
