#!/bin/bash
# Round-3 closing measurements on ONE box.  Everything lands under gpurun_out/ (copy what is judged into profiles/).
set -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout -k 10 300 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r03final -o r03final -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/r03final_bench.log 2>&1 ) || exit 1
echo "trace done"
timeout -k 10 400 python scratch/layer_table.py r03 > gpurun_out/r03_layer_table.log 2>&1 || exit 1
echo "layer table done"
timeout -k 10 200 python scratch/stamp_conv.py 8 > gpurun_out/r03_stamp8.txt 2>&1 || exit 1
timeout -k 10 200 python scratch/stamp_conv.py 4 > gpurun_out/r03_stamp4.txt 2>&1 || exit 1
echo "stamps done"
timeout -k 10 300 python bench.py --workload gan-cls --estimator standin --no-cpu-baseline > gpurun_out/r03_gan_cls_standin.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload gan-cls --estimator resnet101 --no-cpu-baseline > gpurun_out/r03_gan_cls_resnet.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload gan-est --estimator resnet101 --batch 64 --no-cpu-baseline > gpurun_out/r03_gan_est_resnet_b64.json 2>/dev/null || exit 1
echo "gan done"
timeout -k 10 300 python bench.py --fwd-only --graph --batch 16 --size 512 --steps 30 --no-cpu-baseline > gpurun_out/r03_infer512_graph.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --fwd-only --graph --dropout-active --batch 16 --size 512 --steps 30 --no-cpu-baseline > gpurun_out/r03_infer512_graph_dropout.json 2>/dev/null || exit 1
echo "infer done"
