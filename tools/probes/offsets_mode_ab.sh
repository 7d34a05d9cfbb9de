# fresh (proof inside every timed step) vs producer-tagged offsets as SEPARATE processes, interleaved: what the proof costs without the
# in-process ordering of bench.py's own comparison leg
Q="--no-cpu-baseline --no-parity-check --no-box-calibration --no-rccl-selfcheck --no-high-row-check --no-standalone-emb --steps 30 --warmup 5"
for r in 1 2 3; do
  for m in fresh tagged resident; do
    python bench.py $Q --offsets $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ip=d.get('iota_proof') or {}
print('$m', round(d['ms_per_step'],3), 'tagged-leg', ip.get('ms_per_step_with_producer_tagged_offsets'), 'proofs', ip.get('device_proofs_in_timed_region'), 'wait_us', ip.get('host_wait_us_per_step'))"
  done
done
