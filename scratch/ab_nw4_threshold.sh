#!/bin/bash
# Same-box interleaved comparison of the 4-wave threshold of the conv instances (wu_set_option 0: 1 = Cin >= 256, N >= 64 = Cin >= N)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in 1 192 128; do
    timeout -k 10 200 python bench.py --opt 0=$v --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('opt 0=$v', d['ms_per_step'], 'ms (median', d['ms_per_step_median'], ')', d['value'], 'img/s; conv launch', r['avg_launch_ms'], 'ms in-step,', r['single_stream']['avg_launch_ms'], 'single-stream')" || exit 1
  done
done
