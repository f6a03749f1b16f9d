#!/bin/bash
# Same-box A/B of two builds of the library: alternates bench.py runs (A = libfrp_base.so from tools/ab_lib.sh,
# B = the in-tree libfrp.so) and prints ms/step, stage times and conv TFLOP/s of each run.
#   tools/ab_bench.sh [rounds] [extra bench.py flags...]
rounds=${1:-2}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then export FRP_LIB=$root/face-recognition-platform_amd/libfrp_base.so; else unset FRP_LIB; fi
    python bench.py --steps 20 --warmup 3 --cpu-frames 0 --pcie-steps 0 --threshold-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['config']['stage_ms_per_step']
print('$v', 'ms/step %.3f' % d['ms_per_step'], 'det %.3f emb %.3f' % (s['det_conv'], s['emb_conv']), 'conv TF %.1f' % d['roofline']['achieved'])"
  done
done
