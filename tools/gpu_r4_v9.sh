#!/bin/bash
OUT=gpurun_out/r4v9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "lookup_sort or dcn or mlperf_v2_bench" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python tools/probes/cu_mask_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/cu_mask_probe.txt
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration"
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 > $OUT/bench_v2_dcn.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_v2_dot.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_kaggle -o k -- python $GRAFT_REPO_ROOT/bench.py --workload criteo_kaggle --steps 200 --warmup 20 $FLAGS --graph > $GRAFT_REPO_ROOT/$OUT/bench_kaggle_graph_prof.json 2> /dev/null
cd $GRAFT_REPO_ROOT
find $OUT/prof_kaggle -name "*kernel_trace.csv" -delete
timeout 200 python bench.py --workload criteo_kaggle --steps 200 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph.json 2>/dev/null
python - <<PY
import json, csv, glob
for n in ("bench_v2_dcn","bench_v2_dot","bench_kaggle_graph"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.3f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()}, (d.get("parity_check") or {}).get("pass"))
    except Exception as e: print(n, "failed", e)
f=glob.glob("$OUT/prof_kaggle/*kernel_stats.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:40]:
        print("%-90s calls %6s avg %8.2f us tot %9.1f" % (r["Name"].replace("(anonymous namespace)::","")[:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
