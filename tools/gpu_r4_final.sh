#!/bin/bash
# Round-4 build, evidence at HEAD after the fused lookup + interaction became the default forward / backward: the default bench line (both
# baseline legs), rocprofv3 kernel statistics + step trace, the two PMC passes of the same command, the secondary workloads.
# (The whole GPU suite runs separately: tools/gpu_r4_suite.sh.)
OUT=gpurun_out/${1:-r4final}
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench (default)";  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
for a in ("alt_two_kernel_lookup", "alt_mlp_arith", "alt_stream_overlap"): print(a, (d.get(a) or {}).get("ms_per_step"))
c=d.get("cpu_baseline") or {}; print("cpu", c.get("value"), c.get("ms_per_step"), c.get("threads"), c.get("iterations_run"))
s=d.get("stock_gpu_baseline") or {}; print("stock", s.get("value"), s.get("ms_per_step"), s.get("error"))
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_note"])
PY
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap --no-alt-fuse --no-box-calibration"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $FLAGS > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
tr=$(find $OUT/rocprof -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python tools/step_trace.py "$tr" 8 > $OUT/step_trace.txt 2>&1; tail -1 $OUT/step_trace.txt
db=$(find $OUT/rocprof -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py "$db" --out $OUT/rocprof_kernel_stats.md; python tools/rocpd_summary.py "$db" --by-grid --out $OUT/rocprof_kernel_stats_by_grid.md; fi
st=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" $OUT/rocprof_kernel_stats.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete; find $OUT -name "*.db" -size +8M -delete
echo "== secondary workloads"
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dot.json 2> /dev/null
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dcn.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --optimizer rwsadagrad > $OUT/bench_tb_rwsadagrad.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16 > $OUT/bench_tb_bf16.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16x6 > $OUT/bench_tb_bf16x6.json 2> /dev/null
timeout 300 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph.json 2> /dev/null
timeout 300 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS > $OUT/bench_kaggle_eager.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --graph > $OUT/bench_tb_graph.json 2> /dev/null
python - <<PY
import json
for n in ("bench_tb_rwsadagrad","bench_tb_bf16","bench_tb_bf16x6","bench_tb_graph","bench_kaggle_eager","bench_kaggle_graph","bench_mlperf_v2_dot","bench_mlperf_v2_dcn"):
    try:
        d=json.load(open("$OUT/%s.json" % n)); p=d.get("parity_check") or {}
        print("%-28s ms %.3f  update=%s lookup=%s" % (n, d["ms_per_step"], d["config"]["embedding_update"][:20], d["config"].get("embedding_interaction", "")[:12]))
    except Exception as e: print(n, "failed", e)
PY
