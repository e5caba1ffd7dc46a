#!/bin/bash
# a kernel trace of the U-Net step with the bucketed RCCL reducer forced on (world 1, the python interpreter directly
# after "--"): per bucket collective, when its RCCL kernel started / ended relative to the persistent conv and weight-gradient kernels.
#   scratch/rccl_overlap_trace.sh   ->  gpurun_out/r04_rccl_overlap.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_rccl -o rccl -- python3 $root/bench.py --force-ddp --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $root/gpurun_out/rccl_bench.log 2>&1
cd $root && python3 - <<'PY' > gpurun_out/r04_rccl_overlap.txt
import csv, glob, re, collections
fs = glob.glob("gpurun_out/prof_rccl/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    m = re.search(r"(ncclDevKernel\w*|rccl\w*|conv3x3_mfma_v2_kernel|conv3x3_wgrad_v2_kernel|wgrad_reduce_kernel|adain_upcat_\w+|maxpool2_bwd_kernel|conv1x1_tanh_\w+|conv3x3_c3_\w+|multi_tensor_apply_kernel|fused_adam\w*)", n)
    r["short"] = m.group(1) if m else n[:50]
rows.sort(key=lambda r: r["s"])
print("bench line:", open("gpurun_out/rccl_bench.log").read().strip().splitlines()[-1][:300])
coll = [r for r in rows if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
print(f"{len(rows)} kernel dispatches in the trace, {len(coll)} of them RCCL kernels; queues used: {sorted(set(r['Queue_Id'] for r in rows))}")
if not coll:
    print("NO RCCL kernel was dispatched: with one rank the all-reduce of a bucket is a no-op inside RCCL (nothing to exchange), so what a")
    print("collective does while every CU holds a persistent conv workgroup cannot be observed on one GPU.")
else:
    # the last complete step: collectives after the last optimizer kernel but one
    t_end = rows[-1]["e"]
    last = [c for c in coll][-13:]
    t0 = min(c["s"] for c in last) - 9_000_000
    print("last step, RCCL kernels (us relative to the first one's start - 9 ms window):")
    for c in last:
        # what else was running when this collective STARTED and ENDED, and the closest persistent-kernel boundaries
        running_s = [r["short"] for r in rows if r["s"] <= c["s"] < r["e"] and r is not c]
        running_e = [r["short"] for r in rows if r["s"] <= c["e"] < r["e"] and r is not c]
        prev_end = max([r["e"] for r in rows if r["e"] <= c["s"] and "conv3x3" in r["short"]] or [c["s"]])
        print(f"  {c['short'][:40]:40s} queue {c['Queue_Id']} grid {c['Grid_Size_X']:>6s}  start {(c['s'] - t0) / 1e3:9.1f}  dur {(c['e'] - c['s']) / 1e3:7.1f} us   "
              f"{(c['s'] - prev_end) / 1e3:6.1f} us after the last conv/wgrad kernel retired   running at start: {running_s}   at end: {running_e}")
PY
cat $root/gpurun_out/r04_rccl_overlap.txt
