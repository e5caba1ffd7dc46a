#!/bin/bash
# After the packed-FP32 change: the concurrency probes, the GAN determinism probe with the stream overlap on, then the A/B of the step.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 200 python scratch/img3_concurrency_probe.py > gpurun_out/img3_probe.txt 2>&1 || exit 1
grep "stem7x7 fwd  \|nothing" gpurun_out/img3_probe.txt
WU_GAN_OVERLAP=1 timeout -k 10 300 python scratch/gan_determinism_probe.py est 64 5 > gpurun_out/gan_det_overlap.txt 2>&1 || exit 1
grep -v "^run" gpurun_out/gan_det_overlap.txt | cut -c1-130
bash scratch/ab_step.sh 3 --steps 20 --warmup 5 --no-roofline > gpurun_out/ab_packed_step.txt 2>&1 || exit 1
cat gpurun_out/ab_packed_step.txt
