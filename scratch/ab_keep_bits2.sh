#!/bin/bash
# one call: kernel-level bit-equality of the redraw ring kernel vs the stored-bytes ring kernel (existing tests), the net-level probe, then the same-box A/B of the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "adain_upcat or keep_bytes or dropout" > gpurun_out/keep2_pytest.txt 2>&1 || { tail -30 gpurun_out/keep2_pytest.txt; exit 1; }
tail -2 gpurun_out/keep2_pytest.txt
timeout -k 10 300 python scratch/keep_bits_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/keep_bits_probe2.txt || { tail -20 gpurun_out/keep_bits_probe2.txt; exit 1; }
head -2 gpurun_out/keep_bits_probe2.txt; tail -1 gpurun_out/keep_bits_probe2.txt
bash scratch/ab_unet_env.sh WU_KEEP_BITS 3 > gpurun_out/keep_bits_ab2.txt 2>&1 || { tail gpurun_out/keep_bits_ab2.txt; exit 1; }
cat gpurun_out/keep_bits_ab2.txt
