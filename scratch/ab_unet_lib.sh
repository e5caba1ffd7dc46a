#!/bin/bash
# Same-box interleaved A/B of the U-Net training step: the tree's library against scratch/_oldlib/libwu_old.so (scratch/build_baseline_lib.sh <rev>)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}
for r in $(seq $rounds); do
  for v in new old; do
    if [ $v = old ]; then export WU_AB_LIB=$PWD/scratch/_oldlib/libwu_old.so; else unset WU_AB_LIB; fi
    line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1) || exit 1
    echo "$v $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
  done
done
