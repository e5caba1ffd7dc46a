#!/bin/bash
# one call: correctness probe of WU_KEEP_BITS, then the same-box interleaved A/B of the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 300 python scratch/keep_bits_probe.py > gpurun_out/keep_bits_probe.txt 2>&1 || { tail -20 gpurun_out/keep_bits_probe.txt; exit 1; }
tail -5 gpurun_out/keep_bits_probe.txt
bash scratch/ab_unet_env.sh WU_KEEP_BITS 3 > gpurun_out/keep_bits_ab.txt 2>&1 || { tail gpurun_out/keep_bits_ab.txt; exit 1; }
cat gpurun_out/keep_bits_ab.txt
