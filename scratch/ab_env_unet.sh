#!/bin/bash
# Same-box interleaved A/B of the U-Net step under an environment variable:  ab_env_unet.sh VAR VALUE_A VALUE_B [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; a=$2; b=$3; rounds=${4:-3}
for r in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
