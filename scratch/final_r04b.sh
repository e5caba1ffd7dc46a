#!/bin/bash
# Round-4 re-entry closing run on ONE box: the full GPU suite, the default bench line, the rocprofv3 kernel trace of the same command and the
# three PMC passes keyed on the final sources.  Everything lands under gpurun_out/ (scratch/collect_profiles_r04.py copies what is judged).
set -o pipefail
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_pytest.log 2>&1 || { tail -30 gpurun_out/r04b_pytest.log; exit 1; }
tail -3 gpurun_out/r04b_pytest.log
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04b_smoke.log 2>&1 || { tail gpurun_out/r04b_smoke.log; exit 1; }
echo "smoke done"
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || exit 1
cat gpurun_out/r04_bench_default.json
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r04final -o r04final -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/r04final_bench.log 2>&1 ) || exit 1
echo "trace done"
timeout -k 10 400 python scratch/pmc_collect.py r04 > gpurun_out/r04_pmc.log 2>&1 || { tail gpurun_out/r04_pmc.log; exit 1; }
echo "pmc done"
