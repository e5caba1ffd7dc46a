#!/bin/bash
# What does the roofline leg's event bracketing cost the headline number?  bench.py default against --no-roofline, interleaved on one box.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/ab_roofline_overhead.txt
: > $out
for r in 1 2 3; do
  for f in "" "--no-roofline"; do
    line=$(timeout -k 10 200 python bench.py --no-cpu-baseline $f --steps 20 --warmup 5 2>/dev/null | tail -1) || exit 1
    echo "flags='$f' $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "img/s")')" | tee -a $out
  done
done
