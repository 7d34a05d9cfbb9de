#!/bin/bash
# bench.py's whole N = 2 control flow on a 1-GPU box (gloo, both ranks on cuda:0, reduced sizes): NOT a measurement
OUT=gpurun_out/${1:-selftest}; mkdir -p $OUT
export DLRM_BENCH_SELFTEST_GLOO=1
for sync in ddp flat; do
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 2 \
    --batch 8192 --row-cap 200000 --hang-timeout 120 --dense-sync $sync > $OUT/n2_$sync.json 2> $OUT/n2_$sync.err
echo "rc=$? ($sync)"; grep -v "amdgpu.ids\|watchdog [0-9]" $OUT/n2_$sync.err | tail -8 | cut -c1-300
python - <<PY
import json
try:
    d = json.loads(open("$OUT/n2_$sync.json").read().strip().splitlines()[-1])
    print("value %.0f ms %.3f loss %.5f" % (d["value"], d["ms_per_step"], d["final_loss"]), d["config"]["parallelism"])
    for k in ("alt_a2a_pipelined", "alt_a2a_reference", "alt_dense_sync", "collectives", "distributed", "selftest", "incomplete"):
        if k in d: print(" ", k, str(d[k])[:260])
except Exception as e: print("no json", e)
PY
done
