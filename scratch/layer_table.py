"""Per-layer / per-kernel table of the B=32 256x256 bf16 training step, every kernel timed STAND-ALONE (hipEvents on the launch
stream, median of `reps` launches after warm-up): TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak for the convs, GB/s of
ALGORITHMIC bytes against 8 TB/s for the HBM-bound glue.  Writes markdown to gpurun_out/<tag>_layer_table.md.

    python scratch/layer_table.py [tag] [batch] [size] [option=value ...]
"""
import os
import statistics
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import kernels as K
    from wu.layout import empty_nhwc
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    from wu import _lib
    for kv in sys.argv[4:]:                       # kernel-variant switches, e.g. 8=1 (wu_set_option)
        k, v = kv.split("=")
        _lib.call("wu_set_option", int(k), int(v))
    dev = torch.device("cuda:0")
    bf = torch.bfloat16

    def run(fn, reps=7, inner=4):
        """Median over `reps` of (time of `inner` back-to-back launches) / inner: a single launch between two events carries the
        host's launch latency (10-20 us on an idle stream), which back-to-back launches hide behind the running kernel."""
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / inner)
        return statistics.median(ts)

    def act(c, s, relu=False):
        """A synthetic NHWC activation / gradient: U(-1, 1), or -- for the convs whose input IS a ReLU output in the step (every conv
        except the decoders' first, whose upsampled half is an AdaIN / dropout output) -- ReLU(N(0, 1)): half zeros.  Round 4
        (scratch/down12_probe.py, profiles/r04_down12_probe.txt): the same 64 -> 64 @256 forward runs 153-164 us on the latter and
        162-185 us on the former (the matrix pipe's power follows the operand toggling), and 206 us as the FIRST kernel timed in a
        process -- which is how r03's table read 202 us for down1.2 where the in-step PMC pass read 154 us."""
        t = torch.relu(torch.randn((B, s, s, c), device=dev)) if relu else torch.rand((B, s, s, c), device=dev) * 2 - 1
        return t.to(bf).permute(0, 3, 1, 2)

    # warm-up: ~0.3 s of matrix work before the first timed row (the first timings of a process run 10-25 % slow)
    _wx, _wy = act(128, 128), empty_nhwc(B, 128, 128, 128, bf, dev)
    _wf, _ = K.pack_conv3x3((torch.rand((128, 128, 3, 3), device=dev) * 2 - 1) * 0.05, 1)
    for _ in range(1500):
        K.conv3x3(_wx, _wf, None, _wy, 1, 1)
    torch.cuda.synchronize()
    del _wx, _wy, _wf

    rows_conv, rows_glue = [], []
    # the variants the fused graph launches (wu/unet_graph.py): a block's FIRST conv writes the gate bits of its output and its data
    # gradient is ungated; the SECOND conv is plain (or + fused 2x2 max-pool on the encoder) and its data gradient reads the gate bits
    layers = [("down1.2", 64, 64, S, "pool"), ("down2.0", 64, 128, S // 2, "bits"), ("down2.2", 128, 128, S // 2, "pool"),
              ("down3.0", 128, 256, S // 4, "bits"), ("down3.2", 256, 256, S // 4, "pool"), ("down4.0", 256, 512, S // 8, "bits"),
              ("down4.2", 512, 512, S // 8, "plain"), ("up3.0", 768, 256, S // 4, "bits"), ("up3.2", 256, 256, S // 4, "plain"),
              ("up2.0", 384, 128, S // 2, "bits"), ("up2.2", 128, 128, S // 2, "plain"), ("up1.0", 192, 64, S, "bits"), ("up1.2", 64, 64, S, "head")]
    # TWO passes over the list, the second one reported: whatever the table measures first reads 10-25 % slow (r04: down1.2 forward 205 us
    # as the first row against 151-163 us for the same kernel on the same inputs later in the process, warm-up launches of another shape
    # notwithstanding -- profiles/r04_down12_probe.txt); the first pass's numbers are printed to the log for comparison
    for rep in range(2):
      rows_conv.clear()
      tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
      for name, ci, co, s, kind in layers:
          x, gy = act(ci, s, relu=kind != "bits" or name.startswith("down")), act(co, s)
          w = (torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05
          wf, wd = K.pack_conv3x3(w, 1)
          bias = torch.zeros(co, device=dev)
          y, dx = empty_nhwc(B, co, s, s, bf, dev), empty_nhwc(B, ci, s, s, bf, dev)
          fl = 2.0 * B * s * s * 9 * ci * co
          if kind == "pool":
              pl = empty_nhwc(B, co, s // 2, s // 2, bf, dev)
              gbp, sbp = K.gate_bits_alloc(y), K.gate_bits_alloc(y)          # round 4: + gate / arg-max bits, as the training step launches it
              t_f = run(lambda: K.conv3x3_relu_pool_bits(x, wf, bias, y, pl, gbp, sbp))
          elif kind == "bits":
              gbo = K.gate_bits_alloc(y)
              t_f = run(lambda: K.conv3x3_bits(x, wf, bias, y, 1, gate_bits_out=gbo))
          elif kind == "head":                                            # round 4: the 1x1 head + tanh rides in this conv's epilogue
              hw3, hb3 = torch.rand((3, 64), device=dev) - 0.5, torch.zeros(3, device=dev)
              himg = torch.empty((B, 3, s, s), device=dev)
              t_f = run(lambda: K.conv3x3_relu_head(x, wf, bias, y, hw3, hb3, himg))
          else:
              t_f = run(lambda: K.conv3x3(x, wf, bias, y, 1, 1))
          if kind == "bits":
              t_d = run(lambda: K.conv3x3(gy, wd, None, dx))
          else:
              gbi = torch.randint(-2**31, 2**31 - 1, (K.gate_bits_alloc(dx).numel(),), dtype=torch.int32, device=dev)
              t_d = run(lambda: K.conv3x3_bits(gy, wd, None, dx, egate_bits=gbi))
          dw, db = torch.empty_like(w), torch.empty(co, device=dev)
          t_w = run(lambda: K.conv3x3_wgrad(x, gy, dw, db))
          tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
          rows_conv.append((name, f"{ci}->{co} @{s}", fl / 1e9, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6))
          del x, gy, y, dx

      print(f"pass {rep}: fwd {tot['fwd']:.0f} dgrad {tot['dgrad']:.0f} wgrad {tot['wgrad']:.0f} us; first row {rows_conv[0][3]:.1f} / {rows_conv[0][5]:.1f} / {rows_conv[0][7]:.1f}", flush=True)

    # ---- glue ----
    def g(name, t, nbytes):
        rows_glue.append((name, t, nbytes / 1e6, nbytes / t / 1e3))

    for (c, h, cs) in [(128, S // 2, 64), (256, S // 4, 128), (512, S // 8, 256)]:
        x = act(c, h)
        cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, bf, dev)
        gc = act(c + cs, 2 * h)
        ys = torch.rand((B, c), device=dev) + 0.5
        ym = torch.rand((B, c), device=dev)
        st = K.adain_stats(x, 1e-5)
        dx = empty_nhwc(B, c, h, h, bf, dev)
        lo, hi = B * h * h * c * 2, B * 4 * h * h * c * 2
        g(f"adain_stats C={c} @{h}", run(lambda: K.adain_stats(x, 1e-5)), lo)
        g(f"adain_upcat_fwd C={c} {h}->{2 * h} (p=0.3, mask bits)", run(lambda: K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)), lo + hi + hi // 16)
        mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
        g(f"adain_upcat_bwd C={c} {2 * h}->{h} (stage A + fold + apply, gated)", run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1)),
          hi + hi // 16 + lo + lo)          # dy + mask + x read once, dx written (the parked g' round trip and the second read of x are NOT algorithmic)
        del x, cat, gc, dx
    for (c, h) in [(64, S), (128, S // 2), (256, S // 4)]:
        x, gs = act(c, h), act(c, h)
        gy = act(c, h // 2)
        dxx = empty_nhwc(B, c, h, h, bf, dev)
        t = B * h * h * c * 2
        gbp, sbp = K.gate_bits_alloc(x), K.gate_bits_alloc(x)
        gbp.random_(); sbp.random_()
        g(f"maxpool2_bwd_bits C={c} @{h} (+skip sum, ReLU gate; 2 bits per element instead of the tensor)", run(lambda: K.maxpool2_bwd_bits(gbp, sbp, gy, dxx, gs)), 2 * t + t // 4 + t // 8)
        tb = run(lambda: K.maxpool2_bwd(x, gy, dxx, gs, 1))
        print(f"  (tensor-based maxpool2_bwd C={c} @{h}: {tb:.1f} us)", flush=True)
        del x, gs, gy, dxx
    xi = torch.rand((B, 3, S, S), device=dev) * 2 - 1
    w = torch.rand((64, 3, 3, 3), device=dev) - 0.5
    b = torch.zeros(64, device=dev)
    y = empty_nhwc(B, 64, S, S, bf, dev)
    img, a64 = B * 3 * S * S * 4, B * 64 * S * S * 2
    g("conv3x3_c3_fwd 3->64 (first conv)", run(lambda: K.conv3x3_c3(xi, w, b, y, 1, 1, False, 1)), img + a64)
    gy = act(64, S)
    dw, db = torch.zeros((64, 3, 3, 3), device=dev), torch.zeros(64, device=dev)
    g("conv3x3_c3_wgrad 3->64", run(lambda: K.conv3x3_c3_wgrad(xi, gy, dw, db, 1, 1)), img + a64)
    w3, b3 = torch.rand((3, 64), device=dev) - 0.5, torch.zeros(3, device=dev)
    out = torch.empty((B, 3, S, S), device=dev)
    t_head_alone = run(lambda: K.conv1x1_tanh(y, w3, b3, out))       # not in the step any more (fused into up1.2's forward): reported, not summed
    dxx = empty_nhwc(B, 64, S, S, bf, dev)
    dw3, db3 = torch.zeros((3, 64), device=dev), torch.zeros(3, device=dev)
    g("conv1x1_tanh_bwd 64->3 (ReLU-gated dx)", run(lambda: K.conv1x1_tanh_bwd(out, out, y, w3, dxx, dw3, db3, 1)), 2 * img + 2 * a64)

    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    path = os.path.join(root, "gpurun_out", f"{tag}_layer_table.md")
    with open(path, "w") as fh:
        fh.write(f"# {tag}: per-layer kernel table, cUNet {S}x{S} bf16 B={B}, every kernel stand-alone (median of 7 x (4 back-to-back launches / 4), hipEvents)\n\n")
        fh.write("MFMA convs (peak 2500 TFLOP/s dense bf16): forward = conv+bias+ReLU (+fused 2x2 max-pool and gate / arg-max bits on the encoder blocks' second conv; up1.2: + the 64->3 head + tanh from its epilogue); "
                 "dgrad = data gradient as the fused graph launches it (gate bits of the block's first-conv output in the epilogue for *.2, ungated for *.0); wgrad = weight+bias gradient incl. its split-K reducer.\n\n")
        fh.write("| layer | shape | GFLOP | fwd us | fwd TFLOP/s | dgrad us | dgrad TFLOP/s | wgrad us | wgrad TFLOP/s |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows_conv:
            fh.write(f"| {r[0]} | {r[1]} | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.0f} | {r[5]:.1f} | {r[6]:.0f} | {r[7]:.1f} | {r[8]:.0f} |\n")
        flsum = sum(r[2] for r in rows_conv)
        fh.write(f"| **sum** | | {flsum:.0f} | {tot['fwd']:.0f} | {flsum / tot['fwd'] * 1e3:.0f} | {tot['dgrad']:.0f} | {flsum / tot['dgrad'] * 1e3:.0f} | "
                 f"{tot['wgrad']:.0f} | {flsum / tot['wgrad'] * 1e3:.0f} |\n\n")
        fh.write("HBM-bound kernels (peak 8000 GB/s): algorithmic bytes = every operand read once + every result written once.\n\n")
        fh.write("| kernel | us | algorithmic MB | GB/s | % of 8 TB/s |\n|---|---|---|---|---|\n")
        for r in rows_glue:
            fh.write(f"| {r[0]} | {r[1]:.1f} | {r[2]:.0f} | {r[3]:.0f} | {r[3] / 80:.0f} % |\n")
        fh.write(f"| **sum** | {sum(r[1] for r in rows_glue):.0f} | | | |\n")
        fh.write(f"\nThe 64->3 head + tanh forward is part of up1.2's forward launch above (conv3x3_mfma_v2 GATED = 5); the stand-alone head kernel it replaces in the "
                 f"step (fp32 / narrow-image fallback) measures {t_head_alone:.1f} us here.\n")
    print(open(path).read())


if __name__ == "__main__":
    main()
