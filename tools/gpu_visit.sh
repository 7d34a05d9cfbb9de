#!/bin/bash
# One parameterised GPU visit (replaces the per-visit scripts of rounds 2-4):   tools/gpu_visit.sh <out-tag> <stage> [<stage> ...]
# Everything lands under gpurun_out/<out-tag>/ (merged back by gpurun).  Stages:
#   smoke            __graft_entry__.smoke()
#   tests[=EXPR]     pytest -m gpu over tests/ (EXPR = a -k expression; files with FILES="tests/a.py tests/b.py")
#   bench[=ARGS]     python bench.py ARGS            -> bench.json (default line: headline + roofline + cpu_baseline)
#   quick[=ARGS]     bench.py without baselines / parity / calibration, ARGS appended -> quick_<n>.json (repeatable)
#   rocprof          rocprofv3 --kernel-trace --stats of the quick headline + per-kernel tables + one-step trace
#   pmc              FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) folded into pmc_traffic.json
#   pmcm             MFMA-busy / LDS-conflict passes -> pmc_mfma_lds.md
#   secondary        rwsadagrad / bf16 / bf16x6 / graph / Kaggle / MLPerf-v2 dot + dcn lines
#   ab=ENV           alternating A/B of the quick bench: A = no env, B = ENV (e.g. ab=DLRM_FOO=1), two rounds each
#   abx[=ROUNDS]     like ab for any number of configurations: ABX="name:ENV=v ..;name2:.." (empty ENV = HEAD), interleaved ROUNDS times
#   run=CMD          any command (quoted), output to run_<n>.log
OUT=gpurun_out/${1:?out tag}; shift
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
QUICK="--no-cpu-baseline --no-parity-check --no-box-calibration --no-rccl-selfcheck --no-high-row-check --no-full-size-parity --no-reference-region --no-alt-update-in-backward"
n=0
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("  ms %.3f  value %.0f  loss %.5f  roofline %s frac %s" % (d["ms_per_step"], d["value"], d.get("final_loss", float("nan")), (r.get("kernel") or "")[:24], r.get("frac")))
    print("  " + "  ".join("%s %.3f" % (k[:16], v[0]) for k, v in (r.get("by_category") or {}).items() if v[0] > 0.03))
    for k in ("parity_check", "full_size_parity", "high_row_check", "rccl_selfcheck", "iota_proof"):
        if d.get(k) is not None:
            v = d[k]; print("  %s: %s" % (k, {a: v[a] for a in list(v)[:8]} if isinstance(v, dict) else v))
    c = d.get("cpu_baseline") or {}
    if c: print("  cpu_baseline", c.get("value"), c.get("unit"), c.get("cores"), c.get("kind"))
except Exception as e:
    print("  no JSON line:", e)
PY
}
for stage in "$@"; do
  name=${stage%%=*}; arg=""; [ "$stage" != "$name" ] && arg=${stage#*=}
  n=$((n+1))
  case $name in
    smoke) timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "== smoke rc=$?"; tail -1 $OUT/smoke.log ;;
    tests) timeout ${TEST_TIMEOUT:-2400} python -m pytest ${FILES:-tests} -m gpu -x -q ${arg:+-k "$arg"} > $OUT/pytest_$n.log 2>&1; echo "== tests rc=$? ($arg)"; tail -4 $OUT/pytest_$n.log ;;
    bench) timeout 1200 python bench.py $arg > $OUT/bench.json 2> $OUT/bench.err; echo "== bench rc=$? ($arg)"; summ $OUT/bench.json ;;
    quick) timeout 600 python bench.py --steps 20 --warmup 5 $QUICK $arg > $OUT/quick_$n.json 2> $OUT/quick_$n.err; echo "== quick_$n rc=$? ($arg)"; summ $OUT/quick_$n.json ;;
    rocprof)
      ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $ROOT/$OUT/rocprof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 $QUICK $arg > $ROOT/$OUT/rocprof_bench.json 2> $ROOT/$OUT/rocprof.err ); echo "== rocprof rc=$?"
      tr=$(find $OUT/rocprof -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python tools/step_trace.py "$tr" 8 > $OUT/step_trace.txt 2>&1; tail -1 $OUT/step_trace.txt
      db=$(find $OUT/rocprof -name "*.db" | head -1)
      if [ -n "$db" ]; then python tools/rocpd_summary.py "$db" --out $OUT/rocprof_kernel_stats.md; python tools/rocpd_summary.py "$db" --by-grid --out $OUT/rocprof_kernel_stats_by_grid.md; fi
      st=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" $OUT/rocprof_kernel_stats.csv
      find $OUT -name "*kernel_trace.csv" -size +8M -delete; find $OUT -name "*.db" -size +8M -delete ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/$c -o p -- python $ROOT/bench.py --steps 4 --warmup 2 ${QUICK/--no-high-row-check/} --no-high-row-check $arg > $ROOT/$OUT/$c.log 2>&1 ); echo "== pmc $c rc=$?"
      done
      python tools/pmc_fold.py $OUT && python tools/pmc_to_json.py $OUT $OUT/pmc_traffic.json
      find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete ;;
    pmcm)
      for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        tag=$(echo $c | cut -d' ' -f1)
        ( cd /tmp && timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/$tag -o p -- python $ROOT/bench.py --steps 3 --warmup 2 $QUICK --no-standalone-emb --no-kernel-timers $arg > $ROOT/$OUT/$tag.log 2>&1 ); echo "== pmcm $tag rc=$?"
      done
      python tools/pmc_mfma_table.py $OUT > $OUT/pmc_mfma_lds.md; head -14 $OUT/pmc_mfma_lds.md
      find $OUT -name "*.csv" -size +6M -delete ;;
    secondary)
      timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 --no-rccl-selfcheck > $OUT/bench_mlperf_v2_dot.json 2> /dev/null; echo "== v2 dot"; summ $OUT/bench_mlperf_v2_dot.json
      timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 --no-rccl-selfcheck > $OUT/bench_mlperf_v2_dcn.json 2> /dev/null; echo "== v2 dcn"; summ $OUT/bench_mlperf_v2_dcn.json
      for v in "tb_rwsadagrad:--optimizer rwsadagrad" "tb_bf16:--mlp-arith bf16" "tb_bf16x6:--mlp-arith bf16x6" "tb_graph:--graph" \
               "kaggle_graph:--workload criteo_kaggle --steps 300 --warmup 20 --graph" "kaggle_eager:--workload criteo_kaggle --steps 300 --warmup 20"; do
        t=${v%%:*}; a=${v#*:}
        timeout 300 python bench.py --steps 20 --warmup 5 $QUICK $a > $OUT/bench_$t.json 2> /dev/null; echo "== $t"; summ $OUT/bench_$t.json
      done ;;
    ab)
      for r in 1 2; do
        timeout 300 python bench.py --steps 30 --warmup 5 $QUICK $AB_FLAGS > $OUT/A$r.json 2> $OUT/A$r.err; echo "== A$r"; summ $OUT/A$r.json
        env $arg timeout 300 python bench.py --steps 30 --warmup 5 $QUICK $AB_FLAGS > $OUT/B$r.json 2> $OUT/B$r.err; echo "== B$r ($arg)"; summ $OUT/B$r.json
      done ;;
    abx)   # several configurations interleaved: ABX="name:ENV=..;name2:ENV2=.. ENV3=.." (empty ENV = HEAD), arg = rounds (default 2)
      for r in $(seq 1 ${arg:-2}); do
        IFS=';' read -ra cfgs <<< "$ABX"
        for c in "${cfgs[@]}"; do
          t=${c%%:*}; e=${c#*:}
          env $e timeout 300 python bench.py --steps ${ABX_STEPS:-30} --warmup 5 $QUICK $AB_FLAGS > $OUT/$t$r.json 2> $OUT/$t$r.err; echo "== $t$r ($e)"; summ $OUT/$t$r.json
        done
      done ;;
    run) bash -c "$arg" > $OUT/run_$n.log 2>&1; echo "== run rc=$? ($arg)"; tail -${RUN_TAIL:-12} $OUT/run_$n.log ;;
    *) echo "unknown stage $stage" ;;
  esac
done
