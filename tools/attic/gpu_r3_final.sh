#!/bin/bash
# Round-3 build, final visit: full GPU suite, the default bench line (with both baseline legs), rocprofv3 kernel statistics + the two PMC
# passes of the same command, the secondary workloads.  Everything lands in gpurun_out/final/ and is copied into profiles/round3/.
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
{
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
  python -c "import torch,os,psutil;print('torch',torch.__version__,'gpus',torch.cuda.device_count(),'cpus',os.cpu_count(),'ram GB',psutil.virtual_memory().total/1e9)"
} > $OUT/device.log 2>&1
echo "== smoke";  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $OUT/pytest_gpu.log
echo "== bench (default)";  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; grep -v "amdgpu.ids\|Unable to import" $OUT/bench.err | tail -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
    for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
    print("alt", d.get("alt_mlp_arith")); print("overlap", d.get("alt_stream_overlap"))
    c=d.get("cpu_baseline") or {}; print("cpu", c.get("value"), c.get("ms_per_step"), c.get("threads"), c.get("sample","")[:80])
    s=d.get("stock_gpu_baseline") or {}; print("stock", s.get("value"), s.get("ms_per_step"), s.get("error"))
except Exception as e: print("no bench json", e)
PY
echo "== rocprof kernel stats + step trace"
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $FLAGS > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
tr=$(find $OUT/rocprof -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python tools/step_trace.py "$tr" 8 > $OUT/step_trace.txt 2>&1; tail -1 $OUT/step_trace.txt
db=$(find $OUT/rocprof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/rocpd_summary.py "$db" --out $OUT/rocprof_kernel_stats.md; python tools/rocpd_summary.py "$db" --by-grid --out $OUT/rocprof_kernel_stats_by_grid.md
  head -14 $OUT/rocprof_kernel_stats.md | cut -c1-150
fi
st=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" $OUT/rocprof_kernel_stats.csv
echo "== pmc"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete; find $OUT -name "*.db" -size +8M -delete
echo "== secondary workloads"
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --optimizer rwsadagrad > $OUT/bench_tb_rwsadagrad.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16 > $OUT/bench_tb_bf16.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --graph > $OUT/bench_tb_graph.json 2> /dev/null
timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 $FLAGS > $OUT/bench_kaggle_eager.json 2> /dev/null
timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 $FLAGS --graph > $OUT/bench_kaggle_graph.json 2> /dev/null
DLRM_GRAPH_SORTED=1 timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 $FLAGS --graph > $OUT/bench_kaggle_graph_sorted.json 2> /dev/null
timeout 600 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dot.json 2> $OUT/bench_mlperf_v2_dot.err
timeout 600 python bench.py --workload mlperf_v2_multihot --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dcn.json 2> $OUT/bench_mlperf_v2_dcn.err
python - <<PY
import json
for n in ("bench_tb_rwsadagrad","bench_tb_bf16","bench_tb_graph","bench_kaggle_eager","bench_kaggle_graph","bench_kaggle_graph_sorted","bench_mlperf_v2_dot","bench_mlperf_v2_dcn"):
    try:
        d=json.load(open("$OUT/%s.json" % n)); p=d.get("parity_check") or {}
        print("%-28s ms %.3f  update=%s parity=%s" % (n, d["ms_per_step"], d["config"]["embedding_update"][:20], p.get("pass")))
    except Exception as e: print(n, "failed", e)
PY
