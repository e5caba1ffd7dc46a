#!/bin/bash
# Same-box interleaved re-check of kernel-variant switches whose optimum may have drifted: default against one switch flipped at a time
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
  for kv in "0=1" "6=0" "7=0" "4=0" "11=0"; do
    timeout -k 10 200 python bench.py --opt $kv --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('opt $kv', d['ms_per_step'], 'ms (median', d['ms_per_step_median'], ')', d['value'], 'img/s; conv launch', r['avg_launch_ms'], 'ms in-step')" || exit 1
  done
done
