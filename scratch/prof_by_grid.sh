#!/bin/bash
# usage: scratch/prof_by_grid.sh <tag> <name filter> <python script + args>
#   rocprofv3 --kernel-trace of a script; per (kernel name, grid) call count and median / min duration for kernels whose name contains the filter
tag=$1; pat=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
script=$root/$1; shift; rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $script "$@" > $root/gpurun_out/$tag.log 2>&1
cd $root && python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, statistics, collections, re
fs = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)
if not fs: print("no kernel_trace.csv"); sys.exit(1)
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        m = re.search(r"(\w*" + re.escape(sys.argv[2]) + r"\w*)", r["Kernel_Name"])
        d[(m.group(1)[-60:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(f"{k[0]:60s} grid {k[1]:>7s} x {k[2]:>4s} x {k[3]:>3s}  calls {len(v):4d}  median {statistics.median(v):8.1f} us  min {min(v):8.1f} us")
PY
