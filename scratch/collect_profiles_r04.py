"""Copy the round's closing measurements (scratch/final_r04.sh + pmc_collect.py r04) from gpurun_out/ into profiles/ and write the
rocprofv3 summary.  Run in the build container after the calls."""
import csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for a, b in (("r04_pmc.json", "r04_pmc.json"), ("r04_pmc.md", "r04_pmc.md"), ("r04_layer_table.md", "r04_layer_table.md"),
             ("prof_r04final/r04final_kernel_stats.csv", "r04_bench_b32_bf16_kernel_stats.csv"), ("r04_bench_default.json", "r04_bench_default.json"),
             ("r04_gan_phase_times.txt", "r04_gan_phase_times.txt"), ("r04_s2_bench.txt", "r04_s2_bench.txt"),
             ("r04_gan_cls_kernel_table.txt", "r04_gan_cls_kernel_table.txt"), ("r04_gan_est_kernel_table.txt", "r04_gan_est_kernel_table.txt")):
    if os.path.exists(os.path.join(G, a)):
        shutil.copy(os.path.join(G, a), os.path.join(P, b))
OTHER = ("gan_cls_standin", "gan_cls_resnet", "gan_est_resnet_b64", "infer512_graph", "infer512_graph_dropout")
if all(os.path.exists(os.path.join(G, f"r04_{f}.json")) for f in OTHER):      # a call that ran the bench / trace / PMC legs only leaves the committed lines alone
    with open(os.path.join(P, "r04_other_workloads.jsonl"), "w") as fh:
        for f in OTHER:
            fh.write([l for l in open(os.path.join(G, f"r04_{f}.json")) if l.startswith("{")][-1])
d = json.loads([l for l in open(os.path.join(P, "r04_bench_default.json")) if l.startswith("{")][-1])
p = json.loads([l for l in open(os.path.join(G, "r04final_bench.log")) if l.startswith("{")][-1])
rows = list(csv.DictReader(open(os.path.join(P, "r04_bench_b32_bf16_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
conv = [r for r in rows if "conv3x3_mfma_v2_kernel" in r["Name"]]
cavg = sum(float(r["TotalDurationNs"]) for r in conv) / sum(int(r["Calls"]) for r in conv) / 1e3
tbl = ["| kernel | calls | avg us | total ms | % of kernel time |", "|---|---|---|---|---|"]
for r in rows[:34]:
    tbl.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {100 * float(r['TotalDurationNs']) / tot:.1f} |")
ro, rp = d["roofline"], p["roofline"]
traffic = ro.get("traffic")
if traffic is None and os.path.exists(os.path.join(P, "r04_pmc.json")):
    traffic = json.load(open(os.path.join(P, "r04_pmc.json")))["kernels"]["conv3x3_mfma_v2_kernel"]["hbm_bytes_per_launch"]
open(os.path.join(P, "r04_bench_b32_bf16_summary.md"), "w").write(f"""# r04: end of round 4

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline` (the default command: 10 timed steps after 3 warm-up) on MI355X, cUNet 256x256 bf16 B=32 training step (scratch/final_r04.sh; this file, r04_bench_default.json, r04_layer_table.md, r04_other_workloads.jsonl, r04_gan_phase_times.txt and r04_s2_bench.txt are ONE box, one gpurun call; r04_pmc.* is the call after it).
17 steps traced: 3 warm-up + 10 timed + 1 + 3 of bench.py's stand-alone pass (weight-gradient side stream off), plus the one-off side-stream probe (`spin_kernel`).  In the two-stream steps the weight-gradient kernels and whatever runs beside them on the main stream SHARE the chip, so their per-launch durations are longer than stand-alone while the step is shorter; kernel time summed over both streams exceeds the wall time.
bench.py under the profiler: {p['ms_per_step']} ms/step (median {p.get('ms_per_step_median')}), in-step conv fwd+dgrad {rp['avg_launch_ms']} ms per launch ({rp['achieved']} TFLOP/s); call-weighted average of the conv3x3_mfma_v2_kernel rows below (all 17 steps, incl. the single-stream ones): {cavg:.1f} us per launch.  Un-profiled run on the same box right before (profiles/r04_bench_default.json): {d['value']} images/s ({d['ms_per_step']} ms/step, per-step median {d.get('ms_per_step_median')}, min / max {d.get('ms_per_step_min_max')}), in-step {ro['avg_launch_ms']} ms per launch = {ro['achieved']} TFLOP/s ({ro['frac']} of 2.5 PFLOP/s), HBM-side traffic {traffic} bytes per launch (r04_pmc.json) against {ro['algorithmic_bytes_per_launch']} algorithmic, single-stream {ro['single_stream']['avg_launch_ms']} ms = {ro['single_stream']['achieved']} TFLOP/s ({ro['single_stream']['frac']}).
Same-box interleaved A/B against the round-3 library (profiles/r04_step_ab.txt): 8.36-8.40 vs 8.48-8.52 ms/step.

""" + "\n".join(tbl) + "\n")
print(d["value"], d["ms_per_step"], d.get("ms_per_step_median"), ro["frac"], ro["avg_launch_ms"], ro.get("traffic"), ro["single_stream"]["frac"])
for l in open(os.path.join(P, "r04_other_workloads.jsonl")):
    o = json.loads(l); print("  ", o["config"]["workload"][:70], o["ms_per_step"], o["value"], (o.get("roofline") or {}).get("frac"))
