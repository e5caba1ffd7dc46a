#!/bin/bash
# Round-4 closing measurements on ONE box.  Everything lands under gpurun_out/ (scratch/collect_profiles_r04.py copies what is judged into
# profiles/).  Two gpurun calls: this one, then `python scratch/pmc_collect.py r04` (three rocprofv3 --pmc passes).
set -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r04final -o r04final -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/r04final_bench.log 2>&1 ) || exit 1
echo "trace done"
timeout -k 10 400 python scratch/layer_table.py r04 > gpurun_out/r04_layer_table.log 2>&1 || exit 1
echo "layer table done"
timeout -k 10 300 python bench.py --workload gan-cls --estimator standin --no-cpu-baseline > gpurun_out/r04_gan_cls_standin.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload gan-cls --estimator resnet101 --no-cpu-baseline > gpurun_out/r04_gan_cls_resnet.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload gan-est --estimator resnet101 --batch 64 --no-cpu-baseline > gpurun_out/r04_gan_est_resnet_b64.json 2>/dev/null || exit 1
echo "gan done"
timeout -k 10 300 python bench.py --fwd-only --graph --batch 16 --size 512 --steps 30 --no-cpu-baseline > gpurun_out/r04_infer512_graph.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --fwd-only --graph --dropout-active --batch 16 --size 512 --steps 30 --no-cpu-baseline > gpurun_out/r04_infer512_graph_dropout.json 2>/dev/null || exit 1
echo "infer done"
timeout -k 10 200 python scratch/gan_phase_time.py cls 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_gan_phase_times.txt || exit 1
timeout -k 10 200 python scratch/bench_s2.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s2_bench.txt || exit 1
timeout -k 10 300 python scratch/bench_s2_graph.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s2_gather_bench_final.txt || exit 1
timeout -k 10 300 python scratch/pw_vs_blas.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pw3_bench_final.txt || exit 1
echo "gan phases / stride-2 / pointwise done"
scratch/prof_gan_kernels.sh gan-cls 32 gankf > gpurun_out/r04_gan_cls_kernel_table.txt 2>&1 || exit 1
scratch/prof_gan_kernels.sh gan-est 64 gankg > gpurun_out/r04_gan_est_kernel_table.txt 2>&1 || exit 1
echo "gan kernel tables done"
