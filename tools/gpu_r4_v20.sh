#!/bin/bash
# round 4, visit 20: row-wise Adagrad pass 1 with the group-level resolve (keys / values / bags of 64 entries at once, rows from registers)
OUT=gpurun_out/r4v21
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "adagrad or bf16x6_from_planes or bf16x3 or mlperf_v2_bench_configuration" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-box-calibration --no-parity-check"
for kc in 4 2 1; do
DLRM_ADAGRAD_KC=$kc timeout 300 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 $FLAGS > $OUT/bench_v2_dot_kc$kc.json 2>/dev/null
DLRM_ADAGRAD_KC=$kc timeout 300 python bench.py --optimizer rwsadagrad --steps 20 --warmup 5 $FLAGS > $OUT/bench_tb_rwsadagrad_kc$kc.json 2>/dev/null
done
python - <<PY
import json
for n in ("bench_v2_dot_kc4","bench_v2_dot_kc2","bench_v2_dot_kc1","bench_tb_rwsadagrad_kc4","bench_tb_rwsadagrad_kc2","bench_tb_rwsadagrad_kc1"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-26s ms %.3f" % (n, d["ms_per_step"]), {k: v[0] for k, v in d["roofline"]["by_category"].items() if k.startswith("emb")})
    except Exception as e: print(n, "failed", e)
PY
