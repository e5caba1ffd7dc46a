#!/bin/bash
# Same-box interleaved A/B of the U-Net training step: the conv kernel's deferred output stores with cache-policy bits (scratch/_oldlib/libwu_aux<N>.so =
# the tree's objects with conv3x3_mfma_v2.hip compiled -DWU_CONV_STORE_AUX=N: 16 = sc1 write-through, 17 = sc0 sc1, 2 = nt) against the tree's plain stores.
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}
WU_AB_LIB=$PWD/scratch/_oldlib/libwu_aux16.so timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv3x3" -x 2>&1 | tail -2 || exit 1
for r in $(seq $rounds); do
  for v in plain aux16 aux17 aux2; do
    if [ $v = plain ]; then unset WU_AB_LIB; else export WU_AB_LIB=$PWD/scratch/_oldlib/libwu_$v.so; fi
    line=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1) || exit 1
    echo "$v $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
  done
done
