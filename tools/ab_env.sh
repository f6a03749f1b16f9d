#!/bin/bash
# Same-box A/B of ONE build under two environments: alternates bench.py runs with and without `VAR=value` (A = with, B = without)
# and prints ms/step, stage times and conv TFLOP/s of each run.
#   tools/ab_env.sh VAR=value [rounds] [extra bench.py flags...]
kv=$1; shift
rounds=${1:-2}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then pre="env $kv"; else pre=""; fi
    $pre python bench.py --steps 20 --warmup 3 --cpu-frames 0 --pcie-steps 0 --threshold-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['config']['stage_ms_per_step']
print('$v', 'ms/step %.3f' % d['ms_per_step'], 'det %.3f emb %.3f' % (s['det_conv'], s['emb_conv']), 'conv TF %.1f' % d['roofline']['achieved'], 'value %.0f' % d['value'])"
  done
done
