#!/bin/bash
# usage: scratch/prof.sh <tag> <python script + args>   -- rocprofv3 kernel stats of a script, top kernels printed
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
script=$root/$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $script "$@" > $root/gpurun_out/$tag.log 2>&1
cd $root && python3 - "$tag" <<'PY'
import csv, glob, sys
fs = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)
if not fs: print("no kernel_stats.csv"); sys.exit(1)
for r in list(csv.DictReader(open(fs[0])))[:25]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
