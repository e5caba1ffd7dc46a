#!/bin/bash
# Same-box A/B of the WHOLE training step: the tree's library against scratch/_oldlib/libwu_old.so (scratch/build_baseline_lib.sh <rev>),
# interleaved, default bench command (minus the CPU baseline).  usage: scratch/ab_step.sh [rounds] [bench args...]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
rounds=${1:-3}; shift
for i in $(seq 1 $rounds); do
  for v in new old; do
    if [ $v = old ]; then export WU_AB_LIB=$root/scratch/_oldlib/libwu_old.so; else unset WU_AB_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('$v', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'in-step conv ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'single-stream', (r.get('single_stream') or {}).get('avg_launch_ms'))"
  done
done
