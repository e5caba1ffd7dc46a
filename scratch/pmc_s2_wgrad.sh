#!/bin/bash
# SQ / LDS counters of the stride-2 weight-gradient launches of scratch/bench_s2.py: bank conflicts of the transposing LDS reads on a stride-2 halo, waits
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_s2w -o g -- python3 $root/scratch/bench_s2.py 32 > $root/gpurun_out/pmc_s2w.log 2>&1
cd $root && python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_s2w/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for r in csv.DictReader(open(cc)):
    if "wgrad_kernel" not in r["Kernel_Name"]: continue
    rows[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[r["Dispatch_Id"]] = (r["Kernel_Name"][30:90], r["Grid_Size"], r["LDS_Block_Size"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d, c in rows.items():
    g = meta[d]; cnt[g] += 1
    for k, v in c.items(): agg[g][k] += v
print("kernel / grid(threads) / lds     n   wave_cyc  wait_any%  active%  wait_lds%  lds_conflict%of_lds_active  gui_active/8")
for g in sorted(agg, key=lambda x: (x[0], int(x[1]))):
    a = agg[g]; n = cnt[g]; wc = a["SQ_WAVE_CYCLES"] or 1
    print(f"{g[0]:60s} {g[1]:>8s} {g[2]:>7s} {n:4d} {wc/n:10.0f} {100*a['SQ_WAIT_ANY']/wc:9.1f} {100*a['SQ_ACTIVE_INST_ANY']/wc:8.1f} {100*a['SQ_WAIT_INST_LDS']/wc:9.1f} {100*a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_LDS_IDX_ACTIVE'],1):12.1f} {a['GRBM_GUI_ACTIVE']/n/8:14.0f}")
PY
