"""Copy the round's closing measurements (scratch/final_r03.sh + pmc_collect.py, ONE gpurun call) from gpurun_out/ into profiles/ and
write the rocprofv3 summary.  Run in the build container after the call."""
import csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for a, b in (("r03_pmc.json", "r03_pmc.json"), ("r03_pmc.md", "r03_pmc.md"), ("r03_layer_table.md", "r03_layer_table.md"),
             ("prof_r03final/r03final_kernel_stats.csv", "r03_bench_b32_bf16_kernel_stats.csv"), ("r03_bench_default.json", "r03_bench_default.json")):
    shutil.copy(os.path.join(G, a), os.path.join(P, b))
with open(os.path.join(P, "r03_other_workloads.jsonl"), "w") as fh:
    for f in ("gan_cls_standin", "gan_cls_resnet", "gan_est_resnet_b64", "infer512_graph", "infer512_graph_dropout"):
        fh.write([l for l in open(os.path.join(G, f"r03_{f}.json")) if l.startswith("{")][-1])
with open(os.path.join(P, "r03_stamps.txt"), "w") as fh:
    for n, title in ((8, "round 3: carried MFMAs, resident weights"), (4, "round 3: 16x16x32 main loop, LDS-transposed epilogue front")):
        fh.write(f"### scratch/stamp_conv.py {n} ({title})\n" + "".join(l for l in open(os.path.join(G, f"r03_stamp{n}.txt")) if "libdrm" not in l) + "\n")
d = json.loads([l for l in open(os.path.join(P, "r03_bench_default.json")) if l.startswith("{")][-1])
p = json.loads([l for l in open(os.path.join(G, "r03final_bench.log")) if l.startswith("{")][-1])
rows = list(csv.DictReader(open(os.path.join(P, "r03_bench_b32_bf16_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
conv = [r for r in rows if "conv3x3_mfma_v2_kernel" in r["Name"]]
cavg = sum(float(r["TotalDurationNs"]) for r in conv) / sum(int(r["Calls"]) for r in conv) / 1e3
tbl = ["| kernel | calls | avg us | total ms | % of kernel time |", "|---|---|---|---|---|"]
for r in rows[:32]:
    tbl.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |")
ro, rp = d["roofline"], p["roofline"]
pmc = json.load(open(os.path.join(P, "r03_pmc.json")))["kernels"]["conv3x3_mfma_v2_kernel"]
traffic = ro.get("traffic") or pmc["hbm_bytes_per_launch"]        # the PMC passes run after the bench of the same call
open(os.path.join(P, "r03_bench_b32_bf16_summary.md"), "w").write(f"""# r03: end of round 3 -- MFMAs carried across the chunk-top barrier, resident weight slab (Cin = Cout = 64), 16x16x32 main loop on the 4-wave instances

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline` (the default command: 10 timed steps after 3 warm-up) on MI355X, cUNet 256x256 bf16 B=32 training step (scratch/final_r03.sh; this file, r03_bench_default.json, r03_pmc.*, r03_layer_table.md, r03_stamps.txt and r03_other_workloads.jsonl are ONE box, one gpurun call).
17 steps traced: 3 warm-up + 10 timed + 1 + 3 of bench.py's stand-alone pass (weight-gradient side stream off), plus the one-off side-stream probe (`spin_kernel`).  In the two-stream steps the weight-gradient kernels and whatever runs beside them on the main stream SHARE the chip, so their per-launch durations are longer than stand-alone while the step is shorter; kernel time summed over both streams exceeds the wall time.
bench.py under the profiler: {p['ms_per_step']} ms/step, in-step conv fwd+dgrad {rp['avg_launch_ms']} ms per launch ({rp['achieved']} TFLOP/s); call-weighted average of the conv3x3_mfma_v2_kernel rows below (all 17 steps, incl. the single-stream ones): {cavg:.1f} us per launch.  Un-profiled run on the same box right before (profiles/r03_bench_default.json): {d['value']} images/s ({d['ms_per_step']} ms/step), in-step {ro['avg_launch_ms']} ms per launch = {ro['achieved']} TFLOP/s ({ro['frac']} of 2.5 PFLOP/s), HBM-side traffic {traffic} bytes per launch (r03_pmc.json, same call) against {ro['algorithmic_bytes_per_launch']} algorithmic, single-stream {ro['single_stream']['avg_launch_ms']} ms = {ro['single_stream']['achieved']} TFLOP/s ({ro['single_stream']['frac']}).
Boxes of the pool differ by +-3 % (the same code measured 8.65-8.70 ms/step, 0.2024-0.2039 ms per conv launch = 0.408-0.411, on the boxes of the interleaved A/Bs: profiles/r03_step_ab.txt, 8.69 vs 8.74 ms/step and 0.411 vs 0.399 of peak against the round-2 library); only same-box comparisons carry a conclusion.

""" + "\n".join(tbl) + "\n")
print(d["value"], d["ms_per_step"], ro["frac"], ro["avg_launch_ms"], ro.get("traffic"), ro["single_stream"]["frac"])
for l in open(os.path.join(P, "r03_other_workloads.jsonl")):
    o = json.loads(l); print("  ", o["config"]["workload"][:70], o["ms_per_step"], o["value"])
