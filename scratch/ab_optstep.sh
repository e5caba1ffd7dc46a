#!/bin/bash
# same-box interleaved A/B of a bench workload over the values of one wu_set_option key: scratch/ab_optstep.sh KEY "V1 V2 ..." ROUNDS [bench args]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
key=$1; vals=$2; rounds=$3; shift; shift; shift
for i in $(seq 1 $rounds); do for v in $vals; do
  python bench.py --no-cpu-baseline --opt $key=$v "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('option $key=$v', 'ms/step', d['ms_per_step'], 'img/s', d['value'])"
done; done
