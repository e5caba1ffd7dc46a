#!/bin/bash
# Same-box interleaved A/B of the GAN iterations: the tree's library against scratch/_oldlib/libwu_old.so (scratch/build_baseline_lib.sh <rev>)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-2}
for r in $(seq $rounds); do
  for v in new old; do
    if [ $v = old ]; then export WU_AB_LIB=$PWD/scratch/_oldlib/libwu_old.so; else unset WU_AB_LIB; fi
    for wl in "gan-cls --batch 32" "gan-est --batch 64"; do
      line=$(timeout -k 10 200 python bench.py --workload $wl --estimator resnet101 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | tail -1) || exit 1
      echo "$v $wl $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "img/s")')"
    done
  done
done
