#!/bin/bash
# Same-box interleaved A/B of the U-Net training step under an environment switch: ab_unet_env.sh VAR [rounds]   (VAR=0 against VAR=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for v in 0 1; do
    line=$(env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1) || exit 1
    echo "$var=$v $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
  done
done
