#!/bin/bash
# SQ counters of the estimator's pointwise GEMM launches (scratch/bench_pw.py 64): are its waves parked (memory / barrier) or issuing?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_pw -o pw -- python3 $root/scratch/bench_pw.py 64 > $root/gpurun_out/pmc_pw.log 2>&1
cd $root && python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/pmc_pw/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(cc)):
    if "conv1x1_mfma" not in r["Kernel_Name"]: continue
    k = (r["Grid_Size"], r.get("LDS_Block_Size", ""))
    rows[(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[r["Dispatch_Id"]] = r["Grid_Size"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d, c in rows.items():
    g = meta[d]; cnt[g] += 1
    for k, v in c.items(): agg[g][k] += v
print("grid(threads)  n   wave_cyc  wait_any%  wait_inst%  active%  wait_lds%  lds_conflict%of_lds_active  gui_active")
for g in sorted(agg, key=lambda x: int(x)):
    a = agg[g]; n = cnt[g]; wc = a["SQ_WAVE_CYCLES"] or 1
    print(f"{g:>10s} {n:4d} {wc/n:10.0f} {100*a['SQ_WAIT_ANY']/wc:9.1f} {100*a['SQ_WAIT_INST_ANY']/wc:10.1f} {100*a['SQ_ACTIVE_INST_ANY']/wc:8.1f} {100*a['SQ_WAIT_INST_LDS']/wc:9.1f} {100*a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_LDS_IDX_ACTIVE'],1):12.1f} {a['GRBM_GUI_ACTIVE']/n/8:14.0f}")
PY
