#!/bin/bash
# round 4, visit 19: arith "bf16x6" from pre-split planes (gemm_bf16.hip PL = 3): kernel + tower tests, the bf16 kernels it shares code with, per-shape rates, step
OUT=gpurun_out/r4v19
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bf16x3 or bf16x6 or gemm_bf16_phased or weight_bf16_from_stored" > $OUT/pytest_kernels.log 2>&1; echo "kernel tests rc=$?"; tail -4 $OUT/pytest_kernels.log
timeout 600 python tools/bf16x6_gemm_bench.py > $OUT/bf16x6_gemm_bench.txt 2>&1; cat $OUT/bf16x6_gemm_bench.txt | cut -c1-400
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-box-calibration"
timeout 300 python bench.py --mlp-arith bf16x6 --steps 20 --warmup 5 $FLAGS > $OUT/bench_tb_bf16x6.json 2>$OUT/err_x6.txt
DLRM_BF16X6_PLANES=0 timeout 300 python bench.py --mlp-arith bf16x6 --steps 20 --warmup 5 $FLAGS --no-parity-check > $OUT/bench_tb_bf16x6_inloop.json 2>/dev/null
python - <<PY
import json
for n in ("bench_tb_bf16x6","bench_tb_bf16x6_inloop"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-24s ms %.3f parity %s" % (n, d["ms_per_step"], (d.get("parity_check") or {}).get("pass")), {k: v[0] for k, v in d["roofline"]["by_category"].items()})
    except Exception as e: print(n, "failed", e)
PY
tail -5 $OUT/err_x6.txt
