#!/bin/bash
# HBM-side traffic of the MFMA kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass) over one
# training step; writes gpurun_out/pmc_traffic.json (copy to profiles/ to have bench.py report it).
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$c -o pmc -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $root/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
cd $root && python3 - <<'PY'
import csv, glob, json, collections
out = {}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]
        key = "conv3x3_mfma_v2" if "conv3x3_mfma_v2_kernel" in k else "conv3x3_wgrad_v2" if "conv3x3_wgrad_v2_kernel" in k else None
        if key: vals[key][c].append(float(r["Counter_Value"]))
for key, d in vals.items():
    n = len(d["FETCH_SIZE"])
    f, w = sum(d["FETCH_SIZE"]) / n, sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    out[key] = {"launches_sampled": n, "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1` (B=32 256x256 bf16, both "
                        "passes include the warm-up step); gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of wide coalesced reads -> "
                        "bytes = (2*FETCH + WRITE)*1024; average over all launches of the kernel (13 forward + 13 data-gradient convs / 13 weight-gradient "
                        "convs per step); memory-side L2 requests (Infinity-Cache hits are counted)"}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items()}))
PY
