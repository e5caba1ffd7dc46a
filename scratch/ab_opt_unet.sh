#!/bin/bash
# Same-box interleaved A/B of the U-Net step under a wu_set_option switch: ab_opt_unet.sh KEY VALUE_A VALUE_B [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
key=$1; a=$2; b=$3; rounds=${4:-3}
for r in $(seq $rounds); do
  for v in $a $b; do
    timeout -k 10 200 python bench.py --opt $key=$v --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('opt $key=$v', d['ms_per_step'], 'ms (median', d['ms_per_step_median'], ')', d['value'], 'img/s; conv launch', r['avg_launch_ms'], 'ms in-step,', r['single_stream']['avg_launch_ms'], 'single-stream')" || exit 1
  done
done
