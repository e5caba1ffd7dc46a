#!/bin/bash
# kernel-time totals of the GAN iteration with the discriminator trunk as one node (1) / one node per layer (0): same kernels?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  WU_DISC_FUSED=$v rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_ganf$v -o g -- python3 $root/bench.py --workload gan-cls --batch 32 --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $root/gpurun_out/ganf$v.log 2>&1
done
cd $root && python3 - <<'PY'
import csv, glob
tabs = []
for v in (0, 1):
    f = glob.glob(f"gpurun_out/prof_ganf{v}/**/*kernel_stats.csv", recursive=True)[0]
    tabs.append({r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))})
names = sorted(set(tabs[0]) | set(tabs[1]), key=lambda n: -max(tabs[0].get(n, (0, 0))[1], tabs[1].get(n, (0, 0))[1]))
print(f"{'kernel':70s} {'calls 0':>8s} {'ms 0':>9s} {'calls 1':>8s} {'ms 1':>9s} {'delta ms':>9s}")
t0 = t1 = 0
for n in names:
    a, b = tabs[0].get(n, (0, 0.0)), tabs[1].get(n, (0, 0.0))
    t0 += a[1]; t1 += b[1]
    if abs(a[1] - b[1]) > 0.05 or a[0] != b[0]:
        print(f"{n[:70]:70s} {a[0]:8d} {a[1]:9.2f} {b[0]:8d} {b[1]:9.2f} {b[1] - a[1]:+9.2f}")
print(f"total kernel time over 7 iterations: per-layer {t0:.1f} ms, fused {t1:.1f} ms")
PY
