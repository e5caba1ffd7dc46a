"""PMC evidence for the training step (run ON the GPU box): memory-side traffic and matrix-pipe occupancy per kernel and per layer.

    python scratch/pmc_collect.py [tag]

Three rocprofv3 passes over `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline` (B=32, 256x256, bf16), each with
--kernel-trace only (no --stats / tracing domains, as the pool requires), counters in SEPARATE passes as MI355X_MICROARCH.md
prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

    pass 1  --pmc FETCH_SIZE
    pass 2  --pmc WRITE_SIZE
    pass 3  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE

gfx950 corrections (same guide): FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = GRBM_GUI_ACTIVE / 8);
SQ_VALU_MFMA_BUSY_CYCLES counts cycles chip-wide (32 per v_mfma_f32_32x32x16_bf16): matrix-pipe occupancy =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x kernel cycles).

Only the dispatches of the LAST step are used (the second half of each kernel's dispatch list).  Conv launches are labelled with
their layer by dispatch order (13 forward, then 13 data-gradient convs; 13 weight-gradient launches).  Writes
gpurun_out/<tag>_pmc.json and gpurun_out/<tag>_pmc.md; copy them to profiles/.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD = ["down1.2", "down2.0", "down2.2", "down3.0", "down3.2", "down4.0", "down4.2", "up3.0", "up3.2", "up2.0", "up2.2", "up1.0", "up1.2"]
DGRAD = ["up1.2", "up1.0", "up2.2", "up2.0", "up3.2", "up3.0", "down4.2", "down4.0", "down3.2", "down3.0", "down2.2", "down2.0", "down1.2"]
SHAPE = {"down1.2": (64, 64, 256), "down2.0": (64, 128, 128), "down2.2": (128, 128, 128), "down3.0": (128, 256, 64), "down3.2": (256, 256, 64),
         "down4.0": (256, 512, 32), "down4.2": (512, 512, 32), "up3.0": (768, 256, 64), "up3.2": (256, 256, 64), "up2.0": (384, 128, 128),
         "up2.2": (128, 128, 128), "up1.0": (192, 64, 256), "up1.2": (64, 64, 256)}
B = 32


def short(name):
    for key in ("conv3x3_mfma_v2_kernel", "conv3x3_wgrad_v2_kernel", "wgrad_reduce_kernel", "adain_upcat_bwd_tile_kernel", "adain_upcat_bwd_apply_rows_kernel",
                "adain_upcat_bwd_march_kernel", "adain_upcat_bwd_gather_kernel", "adain_upcat_bwd_apply_kernel", "adain_upcat_fwd_march_kernel", "adain_upcat_fwd_kernel", "maxpool2_bwd_kernel", "conv1x1_tanh_bwd_kernel", "conv1x1_tanh_fwd_kernel",
                "conv3x3_c3_wgrad_mfma_kernel", "conv3x3_c3_fwd_mfma_kernel", "adain_stats_kernel", "adain_stats_final_kernel", "adain_style_fwd_kernel",
                "adain_style_bwd_kernel", "thin_fold_kernel", "fold_partials_kernel", "pack_conv3x3_kernel", "multi_tensor_apply_kernel"):
        if key in name:
            return key
    return None


def run_pass(tag, counters):
    out = os.path.join(ROOT, "gpurun_out", f"{tag}_{'_'.join(counters)}")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    log = open(out + ".log", "w")
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=True)
    cc = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    rows = collections.defaultdict(dict)            # dispatch id -> {counter: value}
    for r in csv.DictReader(open(cc)):
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = rows[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = []
    for did, (start, d, name) in dur.items():
        disp.append({"id": did, "start": start, "ns": d, "name": name, "key": short(name), **rows.get(did, {})})
    disp.sort(key=lambda x: x["start"])
    return disp


def last_step(disp, key):
    mine = [d for d in disp if d["key"] == key]
    return mine[len(mine) // 2:]                      # warm-up step + timed step: the second half


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    passes = {"FETCH_SIZE": run_pass(tag, ["FETCH_SIZE"]), "WRITE_SIZE": run_pass(tag, ["WRITE_SIZE"]),
              "MFMA": run_pass(tag, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])}
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu import _build
    try:
        import torch
        cus = torch.cuda.get_device_properties(0).multi_processor_count
    except Exception:
        cus = 256
    res = {"source_hash": _build.source_hash(), "workload": f"bench.py --steps 1 --warmup 1, B={B} 256x256 bf16 training step, last step only",
           "corrections": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads); kernel cycles = GRBM_GUI_ACTIVE/8; "
                          "mfma_occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (4*CUs*kernel cycles); durations are those of the profiled passes", "cus": cus,
           "kernels": {}, "conv_layers": [], "wgrad_layers": []}
    keys = sorted({d["key"] for d in passes["FETCH_SIZE"] if d["key"]})
    for key in keys:
        f, w, m = last_step(passes["FETCH_SIZE"], key), last_step(passes["WRITE_SIZE"], key), last_step(passes["MFMA"], key)
        if not f or len(f) != len(w):
            continue
        n = len(f)
        fetch = sum(d.get("FETCH_SIZE", 0.0) for d in f) / n
        write = sum(d.get("WRITE_SIZE", 0.0) for d in w) / n
        ns = sum(d["ns"] for d in f + w) / (2 * n)
        by = (2 * fetch + write) * 1024
        e = {"launches_per_step": n, "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1), "hbm_bytes_per_launch": int(by),
             "avg_us": round(ns / 1e3, 1), "GBps": round(by / ns, 1), "pct_of_8TBps": round(by / ns / 80.0, 1)}
        if m and len(m) == n:
            busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in m) / n
            gui = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in m) / n
            if gui > 0:
                e.update({"SQ_VALU_MFMA_BUSY_CYCLES": int(busy), "GRBM_GUI_ACTIVE": int(gui), "mfma_occupancy": round(busy / (4 * cus * gui / 8.0), 4),
                          "effective_clock_GHz": round(gui / 8.0 / (sum(d["ns"] for d in m) / n), 3)})
        res["kernels"][key] = e
    # per layer: conv launches of the last step in dispatch order
    for name, labels, dest in (("conv3x3_mfma_v2_kernel", [("fwd", l) for l in FWD] + [("dgrad", l) for l in DGRAD], "conv_layers"),
                               ("conv3x3_wgrad_v2_kernel", [("wgrad", l) for l in DGRAD], "wgrad_layers")):
        f, w, m = (last_step(passes[k], name) for k in ("FETCH_SIZE", "WRITE_SIZE", "MFMA"))
        if not (len(f) == len(w) == len(m) == len(labels)):
            res[dest] = f"unexpected launch count {len(f)}/{len(w)}/{len(m)} vs {len(labels)}"
            continue
        for (kind, layer), df, dw, dm in zip(labels, f, w, m):
            ci, co, s = SHAPE[layer]
            fl = 2.0 * B * s * s * 9 * ci * co
            alg = (B * s * s * (ci + co) + 9 * ci * co) * 2
            by = (2 * df.get("FETCH_SIZE", 0.0) + dw.get("WRITE_SIZE", 0.0)) * 1024
            gui = dm.get("GRBM_GUI_ACTIVE", 0.0)
            busy = dm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            ns = (df["ns"] + dw["ns"] + dm["ns"]) / 3.0
            res[dest].append({"pass": kind, "layer": layer, "shape": f"{ci}->{co} @{s}" if kind != "dgrad" else f"{co}->{ci} @{s}", "us": round(ns / 1e3, 1),
                              "TFLOPs": round(fl / ns / 1e3, 1), "hbm_MB": round(by / 1e6, 1), "algorithmic_MB": round(alg / 1e6, 1),
                              "mfma_occupancy": round(busy / (4 * cus * gui / 8.0), 4) if gui else None,
                              "mfma_count_check": round(busy / 32.0 / (fl / (2.0 * 32 * 32 * 16)), 3) if busy else None,
                              "clock_GHz": round(gui / 8.0 / dm["ns"], 3) if gui else None})
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.md"), "w") as fh:
        fh.write(f"# {tag}: PMC counters per kernel and per layer (rocprofv3 --pmc, separate passes; {res['workload']})\n\n{res['corrections']}.\n"
                 f"library source hash {res['source_hash'][:16]}, {cus} CUs.\n\n")
        fh.write("| kernel | launches/step | avg us | FETCH_SIZE KB | WRITE_SIZE KB | HBM-side MB/launch | GB/s | % of 8 TB/s | MFMA occupancy | clock GHz |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, e in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"]):
            fh.write(f"| {k} | {e['launches_per_step']} | {e['avg_us']} | {e['FETCH_SIZE_KB']} | {e['WRITE_SIZE_KB']} | {e['hbm_bytes_per_launch'] / 1e6:.1f} | {e['GBps']} | "
                     f"{e['pct_of_8TBps']} | {e.get('mfma_occupancy', '')} | {e.get('effective_clock_GHz', '')} |\n")
        for dest, title in (("conv_layers", "conv3x3_mfma_v2_kernel per launch (forward, then data gradient)"), ("wgrad_layers", "conv3x3_wgrad_v2_kernel per launch")):
            fh.write(f"\n## {title}\n\n")
            if isinstance(res[dest], str):
                fh.write(res[dest] + "\n")
                continue
            fh.write("| pass | layer | GEMM shape | us (profiled) | TFLOP/s | HBM-side MB | algorithmic MB | MFMA occupancy | MFMA count / algorithmic | clock GHz |\n|---|---|---|---|---|---|---|---|---|---|\n")
            for r in res[dest]:
                fh.write(f"| {r['pass']} | {r['layer']} | {r['shape']} | {r['us']} | {r['TFLOPs']} | {r['hbm_MB']} | {r['algorithmic_MB']} | {r['mfma_occupancy']} | {r['mfma_count_check']} | {r['clock_GHz']} |\n")
    print(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc.md")).read())


if __name__ == "__main__":
    main()
