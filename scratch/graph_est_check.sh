#!/bin/bash
# The estimator's no-grad pass from a hipGraph: tests, then the GAN iterations and the enqueue / wall split with the replay on and off (same box, interleaved)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x -k "graphed_estimator or estimator_bf16_backward" 2>&1 | tail -3 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_resnet.py -q -x -k "full_size or two_process or gan_step or evaluation" 2>&1 | tail -3 || exit 1
for r in 1 2; do
  for v in 1 0; do
    for w in "gan-cls --batch 32" "gan-est --batch 64"; do
      line=$(WU_GAN_GRAPH_EST=$v timeout -k 10 300 python bench.py --workload $w --estimator resnet101 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1) || exit 1
      echo "graph=$v | $w $(echo $line | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms (median", d.get("ms_per_step_median"), ")", d["value"], "img/s")')"
    done
  done
done
for v in 1 0; do
  echo "== WU_GAN_GRAPH_EST=$v"
  WU_GAN_GRAPH_EST=$v timeout -k 10 200 python scratch/gan_phase_time.py cls 32 2>&1 | grep -v amdgpu.ids || exit 1
done
