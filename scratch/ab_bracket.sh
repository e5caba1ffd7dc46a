#!/bin/bash
# Same-box: what the choice of bracketed families does to the step time and to the dominant kernel's in-step duration.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in 0 1; do
    WU_BENCH_BRACKET_ALL=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bracket_all=$v', d['ms_per_step'], 'ms', d['value'], 'img/s  conv in-step', r['avg_launch_ms'], 'ms frac', r['frac'], ' single-stream', r['single_stream']['frac'])"
  done
  python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('no-roofline  ', d['ms_per_step'], 'ms', d['value'], 'img/s')"
done
