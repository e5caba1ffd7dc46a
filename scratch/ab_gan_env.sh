#!/bin/bash
# Same-box interleaved A/B of the GAN iterations under an environment switch: ab_gan_env.sh VAR [rounds]   (VAR=0 against VAR=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for v in 0 1; do
    for wl in "gan-cls --batch 32" "gan-est --batch 64"; do
      line=$(env $var=$v timeout -k 10 200 python bench.py --workload $wl --estimator resnet101 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1) || exit 1
      echo "$var=$v $wl $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
    done
  done
done
