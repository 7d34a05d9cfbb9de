#!/bin/bash
# Round-3 build, visit 3: the prefetching version of the segmented sort — exactness, timing against rocPRIM, per-kernel trace.
OUT=gpurun_out/v3
mkdir -p $OUT
export TMPDIR=/tmp
echo "== sort tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -p no:cacheprovider -k "lookup_sort or emb_bwd_sgd or adagrad_matches" > $OUT/pytest_sort.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_sort.log
echo "== sort alone"
timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1
DLRM_SORT=rocprim timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1
timeout 120 python tools/sort_bench.py 2048 2>&1 | tail -1
DLRM_SORT=rocprim timeout 120 python tools/sort_bench.py 2048 2>&1 | tail -1
echo "== kernel trace of the own sort"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_sort -o s -- python $GRAFT_REPO_ROOT/tools/sort_bench.py 65536 > $GRAFT_REPO_ROOT/$OUT/rocprof_sort.log 2>&1 ); echo "rocprof rc=$?"
f=$(find $OUT/rocprof_sort -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
find $OUT/rocprof_sort -name "*kernel_trace.csv" -size +4M -delete
echo "== A/B in the step"
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "own:" "rocprim:DLRM_SORT=rocprim" "own_b:" "own_graph:" ; do
  tag=${cfg%%:*}; envs=${cfg#*:}; extra=""
  case $tag in *graph) extra="--graph";; esac
  env $envs timeout 300 python bench.py $AB $extra > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    e = k.get("emb_bwd_sgd") or {}
    print("$tag ms %.3f  emb_bwd %s  update=%s sort=%s" % (d["ms_per_step"], e.get("ms_per_step"), d["config"]["embedding_update"][:30], str(d["config"].get("lookup_sort"))[:40]))
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
for g in "" "--graph"; do
  timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap $g > $OUT/kaggle$g.json 2> $OUT/kaggle$g.err
  python -c "
import json
try:
    d=json.load(open('$OUT/kaggle$g.json')); print('kaggle $g ms %.3f update=%s' % (d['ms_per_step'], d['config']['embedding_update'][:40]))
except Exception as e: print('kaggle $g failed', e)"
done
