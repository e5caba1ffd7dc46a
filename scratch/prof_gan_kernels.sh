#!/bin/bash
# full kernel-time table of one GAN workload (default switches): calls per iteration, average us, ms per iteration
# usage: scratch/prof_gan_kernels.sh [gan-cls|gan-est] [batch] [tag]
root=${GRAFT_REPO_ROOT:-/root/repo}
wl=${1:-gan-cls}; b=${2:-32}; tag=${3:-gank}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o g -- python3 $root/bench.py --workload $wl --batch $b --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $root/gpurun_out/$tag.log 2>&1 || exit 1
cd $root && python3 - $tag <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -r[2])
it = 7
tot = sum(r[2] for r in rows)
print(open(f"gpurun_out/{tag}.log").read().strip().splitlines()[-1][:300])
print(f"{'kernel':90s} {'calls/it':>8s} {'avg us':>8s} {'ms/it':>7s} {'%':>5s}")
for n, c, ms in rows[:45]:
    print(f"{n[:90]:90s} {c / it:8.1f} {ms / c * 1e3:8.1f} {ms / it:7.3f} {100 * ms / tot:5.1f}")
print(f"total kernel time per iteration: {tot / it:.2f} ms over {sum(r[1] for r in rows) / it:.0f} launches")
PY
