#!/bin/bash
# Same-box interleaved A/B of the GAN iterations under a library option: ab_gan_opts.sh "3=0" [rounds]   (WU_SET_OPTIONS unset against the given value)
cd ${GRAFT_REPO_ROOT:-/root/repo}
opts=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for v in default "$opts"; do
    for wl in "gan-cls --batch 32" "gan-est --batch 64"; do
      if [ "$v" = default ]; then unset WU_SET_OPTIONS; else export WU_SET_OPTIONS=$v; fi
      line=$(timeout -k 10 200 python bench.py --workload $wl --estimator resnet101 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1) || exit 1
      echo "options $v | $wl $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
    done
  done
done
