#!/bin/bash
# SQ wait / issue counters of every kernel whose name contains $1, for the script + args that follow:
#   scratch/pmc_kernel.sh adain_upcat_bwd_march scratch/ab_upcat_bwd.py
root=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_k -o k -- python3 $root/$1 "${@:2}" > $root/gpurun_out/pmc_k.log 2>&1
cd $root && python3 - "$pat" <<'PY'
import csv, glob, collections, sys
cc = glob.glob("gpurun_out/pmc_k/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for r in csv.DictReader(open(cc)):
    if sys.argv[1] not in r["Kernel_Name"]: continue
    rows[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = (r["Kernel_Name"][:70], r["Grid_Size"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d, c in rows.items():
    cnt[meta[d]] += 1
    for k, v in c.items(): agg[meta[d]][k] += v
for g in agg:
    a = agg[g]; n = cnt[g]; wc = a["SQ_WAVE_CYCLES"] or 1
    print(f"{g[0]} grid {g[1]} x{n}: wait_any {100*a['SQ_WAIT_ANY']/wc:.1f}%  wait_inst {100*a['SQ_WAIT_INST_ANY']/wc:.1f}%  active {100*a['SQ_ACTIVE_INST_ANY']/wc:.1f}%  "
          f"active_valu {100*a['SQ_ACTIVE_INST_VALU']/wc:.1f}%  active_vmem {100*a['SQ_ACTIVE_INST_VMEM']/wc:.1f}%  valu insts/wave-cycle {a['SQ_INSTS_VALU']/wc:.3f}  kernel cycles {a['GRBM_GUI_ACTIVE']/n/8:.0f}")
PY
