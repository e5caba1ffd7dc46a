#!/bin/bash
# same-box interleaved A/B of the whole training step for one environment switch: scratch/ab_env.sh VAR [rounds] [bench args...]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
var=$1; rounds=${2:-3}; shift; shift
for i in $(seq 1 $rounds); do
  for v in 0 1; do
    env $var=$v python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$var=$v', 'ms/step', d['ms_per_step'], 'img/s', d['value'], 'in-step conv ms', r.get('avg_launch_ms'))"
  done
done
