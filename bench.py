#!/usr/bin/env python3
"""Headline benchmark: images/sec, 256x256 cUNet training step (forward + backward [+ gradient all-reduce]
+ Adam) on N MI355X, bf16, synthetic data (BASELINE.json configs[1], B=32 per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (B per GPU fixed).  Rank 0 prints ONE JSON line with the metric, the
roofline of the dominant kernel (timed live with hipEvents on the launch stream inside the timed region)
and, at N=1, the CPU baseline (the oracle restatement, oracle/cunet_ref.py, timed on the host cores on a
bounded sample: B=4 256x256 fp32 fwd+bwd, 4 iterations).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "weather-unet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_FWD_BWD_256 = 254.4      # conv MAC*2, fwd + dgrad + wgrad, per 256x256 image (SURVEY.md 8d / BASELINE.md 3)
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PMC_FILE = "r04_pmc.json"      # scratch/pmc_collect.py output for the current kernels (see roofline.traffic)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true", help="inference forward (eval mode) instead of the training step")
    ap.add_argument("--graph", action="store_true", help="with --fwd-only: replay a hipGraph-captured forward (configs[4])")
    ap.add_argument("--dropout-active", action="store_true",
                    help="with --fwd-only: keep the generator in train mode (Dropout(0.3) active), as the reference's inference scripts do "
                         "(inference/inf_transfer_c.py:88-96 never calls .eval()); with --graph every replay draws a new mask")
    ap.add_argument("--force-ddp", action="store_true", help="initialise RCCL and use the bucketed reducer even with one rank (test hook)")
    ap.add_argument("--estimator", default="resnet101", choices=["resnet101", "standin"],
                    help="gan workloads: the frozen estimator in the loop -- the ResNet-101 the reference uses (classifier.py:106, "
                         "estimator.py:143; random-init weights) or the small stand-in that keeps it out of the step time")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="kernel-variant switch for A/B runs (wu_set_option; defaults are the production choices)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU): every rank joins, rank 0 prints {\"launch_check\": world}; "
                         "exercises the --gpus N self-launch path on a CPU box")
    ap.add_argument("--workload", default="unet", choices=["unet", "gan-cls", "gan-est"],
                    help="unet: cUNet fwd+bwd (the headline metric); gan-cls / gan-est: one full GAN iteration "
                         "(D update + G update) of t_cls_train.py / t_est_train.py (configs[2] / configs[3])")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    args>` as a CHILD process (this parent never touches the GPU and never exec()s) and pass its output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(size, iters=3, batch=4):
    """The oracle (kind: 'port' = stock-PyTorch CPU restatement, bit-identical to the reference, oracle/cunet_ref.py)
    timed on this box's host cores: fwd+bwd of the same loss, fp32 (B=4, 1 warm-up, median of 3)."""
    import statistics
    from oracle import cunet_ref as O
    threads = torch.get_num_threads()
    p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(5, 0).items()}
    x, c = O.make_inputs(batch, size, 5, 0, False)
    times = []
    for it in range(iters + 1):
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        O.bench_loss(O.cunet_forward(p, x, c), x).backward()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times[1:])
    return {"value": round(batch / med, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle fwd+bwd (eval-mode dropout), B={batch} {size}x{size} fp32, median of {iters} after 1 warm-up "
                      f"({med:.2f} s/iter), os.cpu_count()={os.cpu_count()}"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (one fresh process per GPU through the stock
            # launcher) BEFORE this process makes any GPU call, relay rank 0's JSON line, exit with the launcher's code
            raise SystemExit(self_launch(a.gpus))
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": int(t.item()), "n_gpus": world}), flush=True)
        dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_ddp = world > 1 or a.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import cunet
    import ops
    from wu import _lib
    from wu.ddp import GradBucketReducer, ready_order

    for kv in a.opt:
        k, v = kv.split("=")
        if k == "gate_bits":                   # host-side A/B switch: ReLU gates as bits (default) or as tensors
            from wu import unet_graph as _UG
            _UG.GATE_BITS = bool(int(v))
            continue
        _lib.call("wu_set_option", int(k), int(v))
    torch.manual_seed(0)                       # same random-init weights on every rank
    net = cunet.Conditional_UNet(5, precision=a.precision).to(dev)
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)      # a different synthetic shard per rank
    x = (torch.rand((a.batch, 3, a.size, a.size), generator=g) * 2 - 1).to(dev)
    c = torch.eye(5)[(torch.arange(a.batch) + rank * a.batch) % 5].to(dev)

    graphed = None
    gan = None
    if a.workload != "unet":
        from wu.train_step import WeatherTransferStep
        est = None
        if a.estimator == "resnet101":
            from wu.resnet import ResNet101Estimator
            torch.manual_seed(1)
            est = ResNet101Estimator(5, precision=a.precision)
        gan = WeatherTransferStep(5, mode=a.workload[4:], precision=a.precision, device=dev, ddp=use_ddp, seed=0, estimator=est)
        x_rand = (torch.rand((a.batch, 3, a.size, a.size), generator=g) * 2 - 1).to(dev)
        reducer = opt = None
    elif a.fwd_only:
        net.train(a.dropout_active)
        reducer = opt = None
        if a.graph:
            from wu.graph_infer import GraphedUNet
            graphed = GraphedUNet(net, a.batch, a.size)
            graphed.x.copy_(x)
            graphed.c.copy_(c)
    else:
        net.train()                            # Dropout(0.3) active, as in training (cunet.py:28)
        params = list(net.parameters())
        # buckets in gradient-ready order, filled from inside the fused backward.  Layout from the stand-in measurements (profiles/r04_fake_collective.txt): ONE
        # big bucket (29.4 MB, complete after down3.0's weight gradient, 1.2 ms before the end of the backward) + a tail of the last-ready 1.1 MB, whose collective is
        # the only exposed one -- every overlapped collective costs about a third of its duration, so fewer is better as long as the last big one still fits
        reducer = GradBucketReducer(ready_order(net), bucket_mb=32.0, ready_order=True, tail_mb=2.0).attach(net) if use_ddp else None
        opt = torch.optim.Adam(params, lr=1e-4, betas=(0.0, 0.999), weight_decay=1e-4 / 20, fused=True)   # t_cls_train.py:184

    def step():
        if gan is not None:
            return gan.step(x, x_rand)
        if a.fwd_only:
            if graphed is not None:
                return graphed.replay()
            with torch.no_grad():
                return net(x, c)
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        out = net(x, c)
        loss = ops.l1_loss(out, x)                 # mean|G(x,c) - x| (reference ops.py:22-24)
        loss.backward()
        if reducer is not None:
            reducer.finalize()
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    from wu import unet_graph as UG
    UG.prepare_side_stream(dev)                # the one-off side-stream probe must not land in a timed step (warm-up may be 0)
    for _ in range(a.warmup):
        step()
    families = [_lib.FAM_CONV_FWD, _lib.FAM_WGRAD, _lib.FAM_CONV_DGRAD]
    if gan is not None:       # discriminator stride-2 convs and the estimator's pointwise convs
        families += [_lib.FAM_CONV_S2, _lib.FAM_WGRAD_S2, _lib.FAM_CONV1X1]
    do_roof = (not a.no_roofline) and rank == 0
    pre_stats = None
    families_timed = families
    if not a.no_roofline and gan is not None:
        # GAN workloads launch ~550 instrumented kernels per iteration and the hipEvent pair around each costs ~5 us of stream time
        # (3.5 ms of a 31 ms iteration).  One extra, untimed, fully instrumented iteration finds the dominant family and gives the
        # other families' figures; the timed region then brackets the dominant family only (every one of its launches).  Every rank
        # runs the extra iteration (it contains the gradient all-reduce); rank 0 alone instruments it.  The U-Net step keeps all three
        # families bracketed: its 13 weight-gradient pairs sit on the side stream, off the critical path -- leaving them out changed
        # neither the step time nor the overhead (0.18 ms against --no-roofline either way, profiles/r03_roofline_overhead.txt).
        if do_roof:
            _lib.prof_begin(families, 1024 + 64)
        step()
        torch.cuda.synchronize()
        if do_roof:
            pre_stats = {f: _lib.prof_query(f) for f in families}
            _lib.prof_end()
            families_timed = [max(pre_stats, key=lambda f: pre_stats[f]["ms"])]
    if do_roof:
        _lib.prof_begin(families_timed, (64 if gan is None else 1024) * a.steps + 64)
    # one timing-only event per step boundary (recorded on the main stream, no synchronisation): the per-step MEDIAN reported next to
    # the mean -- a single DVFS excursion moves the mean of a 0.1-0.2 s timed region by a few per cent, not the median
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        last = step()
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    step_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if use_ddp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roof = None
    measured_flops = None
    if do_roof:
        stats = {f: _lib.prof_query(f) for f in families_timed}
        if pre_stats is not None:           # the families not bracketed in the timed region: one untimed iteration, scaled to the run
            for f in families:
                if f not in stats:
                    stats[f] = {k: v * a.steps for k, v in pre_stats[f].items()}
        measured_flops = sum(st["flops"] for st in stats.values())     # algorithmic conv FLOPs of the launches in the timed region
        _lib.prof_end()
        graph_note = None
        if graphed is not None:
            # a replayed hipGraph contains no event brackets (they cannot be captured per launch): the same forward is run EAGERLY three
            # more times after the timed region with every conv launch bracketed, and the roofline block is computed from those
            _lib.prof_begin(families, 64 * 3 + 64)
            with torch.no_grad():
                for _ in range(3):
                    net(x, c)
            torch.cuda.synchronize()
            stats = {f: _lib.prof_query(f) for f in families}
            _lib.prof_end()
            graph_note = "3 eager forwards after the timed graph replays, every launch of the listed families bracketed (a replayed graph has no per-launch events)"
        # In the training step the weight-gradient kernels run on a second HIP stream beside the data-gradient chain (faster
        # step), so the per-launch durations above include the time a kernel shares the chip.  A short extra pass with that
        # overlap switched off gives the kernels' stand-alone rate, reported next to the in-step figure.
        iso = None
        if gan is None and not a.fwd_only and world == 1:     # extra steps contain collectives: single-rank runs only
            from wu import unet_graph as UG
            if UG.SIDE_STREAM_WGRAD:
                UG.SIDE_STREAM_WGRAD = False
                step(); torch.cuda.synchronize()
                _lib.prof_begin(families, 64 * 3 + 64)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                iso = {f: _lib.prof_query(f) for f in families}
                _lib.prof_end()
                UG.SIDE_STREAM_WGRAD = True
        # GAN workloads: the passes of an iteration overlap on several streams (D beside G / the estimator, weight gradients beside data
        # gradients), so a launch's event pair above also spans the time it shares the chip or waits behind another stream's kernels.
        # One more untimed iteration with every overlap switched OFF gives each family's rate when its kernels run one at a time.
        seq = None
        if gan is not None and world == 1:
            from wu import unet_graph as UG, train_step as TS
            saved = (UG.SIDE_STREAM_WGRAD, TS.OVERLAP_D_WITH_ESTIMATOR, TS.OVERLAP_D_PASSES)
            UG.SIDE_STREAM_WGRAD, TS.OVERLAP_D_WITH_ESTIMATOR, TS.OVERLAP_D_PASSES = False, False, False
            try:
                step(); torch.cuda.synchronize()
                _lib.prof_begin(families, 1024 + 64)
                step(); torch.cuda.synchronize()
                seq = {f: _lib.prof_query(f) for f in families}
                _lib.prof_end()
            finally:
                UG.SIDE_STREAM_WGRAD, TS.OVERLAP_D_WITH_ESTIMATOR, TS.OVERLAP_D_PASSES = saved
        dom = max(stats, key=lambda f: stats[f]["ms"])
        s = stats[dom]
        peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
        # HBM-side bytes per launch from the PMC counters: collected with rocprofv3 in separate --pmc passes (scratch/pmc_collect.py,
        # gfx950 correction applied) and committed under profiles/ -- bench.py cannot run the profiler on itself.  The file records
        # the content hash of the kernel sources it was measured on; a library built from other sources gets traffic = null.
        traffic, traffic_note = None, None
        try:
            from wu import _build
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as fh:
                pmc = json.load(fh)
            key = {_lib.FAM_CONV_FWD: "conv3x3_mfma_v2_kernel", _lib.FAM_WGRAD: "conv3x3_wgrad_v2_kernel"}.get(dom)
            same_workload = a.precision == "bf16" and a.batch == 32 and a.size == 256 and a.workload == "unet" and not a.fwd_only
            if pmc.get("source_hash") != _build.source_hash():
                traffic_note = f"profiles/{PMC_FILE} was measured on other kernel sources ({str(pmc.get('source_hash'))[:12]} vs {_build.source_hash()[:12]}): not reported"
            elif key in pmc.get("kernels", {}) and same_workload:
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
                traffic_note = f"profiles/{PMC_FILE}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, same kernel sources ({_build.source_hash()[:12]})"
        except (OSError, ValueError, KeyError):
            traffic = None
        if s["launches"]:
            ach = s["flops"] / (s["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_note, "kernel": _lib.FAMILY_KERNEL[dom], "launches": s["launches"],
                    "algorithmic_bytes_per_launch": round(s["bytes"] / s["launches"]),
                    "avg_launch_ms": round(s["ms"] / s["launches"], 4),
                    "algorithmic_gflop_per_launch": round(s["flops"] / s["launches"] / 1e9, 3),
                    "share_of_step": (round(s["ms"] / 3 / (dt / a.steps * 1e3), 3) if graph_note is not None else round(s["ms"] / (dt * 1e3), 3)),
                    "instrumented": (graph_note if graph_note is not None else
                                     "dominant family bracketed in the timed region; other_kernels from one untimed fully instrumented iteration"
                                     if pre_stats is not None else "every launch of the listed families bracketed in the timed region"),
                    "single_stream": (None if iso is None or not iso[dom]["launches"] else
                                      {"achieved": round(iso[dom]["flops"] / (iso[dom]["ms"] * 1e-3) / 1e12, 2),
                                       "frac": round(iso[dom]["flops"] / (iso[dom]["ms"] * 1e-3) / 1e12 / peak, 4),
                                       "avg_launch_ms": round(iso[dom]["ms"] / iso[dom]["launches"], 4),
                                       "note": "same kernels with the weight-gradient side stream off (3 extra steps after the timed region)"}),
                    # the estimator's pointwise GEMMs are HBM-bound (K = 64 ... 2048 at 2-16 k rows: 30-110 flop per algorithmic
                    # byte): their roofline is bandwidth, so they are reported in GB/s of algorithmic bytes as well
                    "other_kernels": {_lib.FAMILY_KERNEL[f]: dict({"ms_per_step": round(stats[f]["ms"] / a.steps, 3),
                                                                   "TFLOP/s": round(stats[f]["flops"] / max(stats[f]["ms"], 1e-9) / 1e9, 1)},
                                                                  **({"launches_per_step": round(stats[f]["launches"] / a.steps, 1),
                                                                      "GB/s": round(stats[f]["bytes"] / max(stats[f]["ms"], 1e-9) / 1e6, 1),
                                                                      "frac_of_8TBps": round(stats[f]["bytes"] / max(stats[f]["ms"], 1e-9) / 1e6 / 8000.0, 4)}
                                                                     if f == _lib.FAM_CONV1X1 else {}))
                                      for f in families}}
            if seq is not None:
                roof["other_kernels_sequential"] = dict(
                    {"note": "one untimed iteration with every overlap off (one stream): each launch's event pair spans that launch alone (+ ~5 us of event records)"},
                    **{_lib.FAMILY_KERNEL[f]: dict({"ms_per_step": round(seq[f]["ms"], 3), "launches_per_step": seq[f]["launches"],
                                                    "TFLOP/s": round(seq[f]["flops"] / max(seq[f]["ms"], 1e-9) / 1e9, 1)},
                                                   **({"GB/s": round(seq[f]["bytes"] / max(seq[f]["ms"], 1e-9) / 1e6, 1),
                                                       "frac_of_8TBps": round(seq[f]["bytes"] / max(seq[f]["ms"], 1e-9) / 1e6 / 8000.0, 4)}
                                                      if f == _lib.FAM_CONV1X1 else {}))
                       for f in families if seq[f]["launches"]})

    if rank == 0:
        imgs = a.batch * world * a.steps
        value = imgs / dt
        res = {
            "metric": ("images/sec GAN iteration (D update + G update)" if gan is not None else
                       "images/sec 256x256 cUNet fwd+bwd" if not a.fwd_only else "images/sec cUNet forward (eval)"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_step_median": round(step_median, 3),
            "ms_per_step_min_max": [round(per_step[0], 3), round(per_step[-1], 3)], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": (f"{'t_' + a.workload[4:] + '_train GAN loop (cUNet + SNDisc + estimator)' if gan is not None else 'cUNet'} "
                                    f"{a.size}x{a.size} {a.precision} B={a.batch}/GPU, "
                                    + (f"estimator = {'frozen ResNet-101 (random-init)' if a.estimator == 'resnet101' else 'small stand-in'}; "
                                       "one estimator pass over cat(rand_images, images) (the reference's three no-grad calls), D update (2 D fwd+bwd, 1 G fwd) + G update (G fwd+bwd, D fwd + data-grad, estimator fwd + data-grad), 2x fused Adam"
                                       if gan is not None else
                                       (("forward only, Dropout(0.3) active" if a.dropout_active else "forward only (eval)") + (", hipGraph replay" if a.graph else "")) if a.fwd_only else
                                       "training step: fwd + bwd (dgrad+wgrad) + grad all-reduce + fused Adam; dropout p=0.3 on; "
                                       "random-init weights, 5-class one-hot, loss mean|G(x,c)-x|")),
                       "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       # conv MAC*2 actually launched in the timed region (3x3 / stride-2 / pointwise families, summed by the launchers) when
                       # the roofline leg ran on this rank alone; else the analytic per-image figure of the cUNet
                       "algorithmic_tflops": (round(measured_flops * world / dt / 1e12, 1) if measured_flops and gan is not None else
                                              round(value * (GFLOP_FWD_BWD_256 if not a.fwd_only else GFLOP_FWD_BWD_256 / 3)
                                                    * (a.size / 256) ** 2 / 1e3, 1))},
        }
        if roof is not None:
            res["roofline"] = roof
        if _lib.LOADED_PATH and os.path.realpath(_lib.LOADED_PATH) != os.path.realpath(_lib.LIB_PATH):
            res["library"] = _lib.LOADED_PATH              # WU_AB_LIB: an A/B run on another build, never a headline number
        if world == 1 and not a.no_cpu_baseline and not a.fwd_only and gan is None:
            res["cpu_baseline"] = cpu_baseline(a.size)
        print(json.dumps(res), flush=True)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
