#!/bin/bash
# A/B: does the per-launch hipEvent instrumentation of the roofline leg cost step time?
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/roofline on  /'
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --no-roofline | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/roofline off /'
done
