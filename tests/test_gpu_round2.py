"""Round-2 GPU parity depth (VERDICT r1 items 1, 3, 8 and the advisor's findings):
 * bf16 gradients of the HIP path against the oracle's bf16-EMULATING mode (same graph, operands rounded where the kernels
   store bf16): separates "precision mode" from "kernel error" -- dconv_up1.* / conv_last.* cosine >= 0.999, every one of the 36
   gradients no further from the emulation than 2x the distance between two emulations that differ only in summation order
   (the stage-by-stage check with the upstream gradient held fixed, cos >= 0.99999 everywhere, is test_gpu_round3.py);
 * the train-mode fixture captured from the reference (its own nn.Dropout masks) compared with the HIP path directly, through
   the injected-mask entry;
 * BASELINE configs[4] at its real size: 512x512, B=16 -- eval determinism, bf16 production kernels vs the fp32 kernels within
   5e-2, hipGraph replay vs eager bit-equal; dropout-ACTIVE graphs (the reference's inference never calls .eval());
 * the packed-operand cache under spectral norm across optimizer steps (advisor, high);
 * the reducer's collective path on a real RCCL process group;
 * the batched evaluation() against the oracle's loop.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cunet_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = {"fp32": 1e-3, "bf16": 5e-2}


def _make_g(nc, seed, precision):
    import cunet
    net = cunet.Conditional_UNet(nc, precision=precision)
    net.load_state_dict(O.make_cunet_params(nc, seed), strict=True)
    return net.to(DEV)


def _make_d(nc, seed, precision):
    import disc
    net = disc.SNDisc(nc, precision=precision)
    net.load_state_dict(O.make_sndisc_params(nc, seed), strict=True)
    return net.to(DEV)


def _cos_rel(a, b):
    a, b = a.detach().cpu().reshape(-1).double(), b.detach().cpu().reshape(-1).double()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item(), ((a - b).norm() / (b.norm() + 1e-30)).item()


# ---------------------------------------------------------------------------------------------------------------------------
# bf16 gradient fidelity
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,train", [((2, 64, 64), False), ((2, 48, 40), False), ((2, 64, 64), True), ((1, 128, 128), True)])
def test_bf16_gradients_vs_emulating_oracle(shape, train):
    """HIP bf16 forward + all 36 parameter gradients against oracle.cunet_forward(emulate_bf16=True): the reference's graph with
    every tensor the kernels park in HBM as bf16 rounded at that point (and its gradient likewise, through autograd).  What is
    left between the two is fp32 summation order.

    Finding (round 2): that is NOT a small difference for the deep layers.  Two CPU emulations of the same bf16 graph that differ
    only in the order their convs visit the input channels (``sum_order``) agree to cosine 0.9998 on dconv_up1.*, but only
    0.96-0.98 from dconv_up2 inwards: the instance-norm backward removes the mean / x-hat components of a gradient dominated by
    those components, so last-bit bf16 flips of that gradient are amplified.  The bf16 gradient is therefore pinned as:
      * forward max-abs vs the emulation <= 1e-2 (and <= 5e-2 vs the fp32 oracle, north_star's tolerance);
      * the layers downstream of the last AdaIN (dconv_up1.*, conv_last.*), which no amplification has touched: cosine >= 0.999,
        relative L2 error <= 5e-2;
      * every one of the 36 gradients: the HIP result is no further from the emulation than 2x the distance between the two
        emulations (+ 2e-3): the kernels sit inside the spread of the precision mode itself.
    """
    from wu import functional as WF
    n, h, w = shape
    nc, seed = 5, 13
    net = _make_g(nc, seed, "bf16")
    net.train(train)
    net.dropout_seed = 21
    r = O._rng("emu", seed)
    x = torch.from_numpy(r.uniform(-1, 1, size=(n, 3, h, w)).astype(np.float32))
    c = torch.softmax(torch.from_numpy(r.standard_normal((n, nc)).astype(np.float32)), 1)
    masks = None
    if train:
        masks = [WF.dropout_mask(n, ch, h // d, w // d, 0.3, (21 * 4 + k) & 0x7FFFFFFFFFFFFFFF, torch.device(DEV)).float().cpu()
                 for k, ch, d in ((3, 512, 4), (2, 256, 2), (1, 128, 1))]
    xd = x.to(DEV)
    out = net(xd, c.to(DEV))
    torch.mean(torch.abs(out - xd)).backward()
    res = {}
    for tag, emu, order in (("emu", True, None), ("emu2", True, 1), ("fp32", False, None)):
        p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
        ref = O.cunet_forward(p, x, c, masks, emulate_bf16=emu, sum_order=order)
        O.bench_loss(ref, x).backward()
        res[tag] = (ref.detach(), p)
    err = (out.detach().cpu() - res["emu"][0]).abs().max().item()
    err32 = (out.detach().cpu() - res["fp32"][0]).abs().max().item()
    err_ee = (res["emu"][0] - res["emu2"][0]).abs().max().item()
    print(f"bf16 {shape} train={train}: forward max-abs vs emulating oracle {err:.3e}, vs fp32 oracle {err32:.3e}; emulation vs re-ordered emulation {err_ee:.3e}")
    assert err <= 1e-2 and err32 <= FWD_TOL["bf16"]
    checked, bad = 0, []
    for k, prm in net.named_parameters():
        if k.endswith("emb.weight"):
            assert prm.grad is None
            continue
        cos, rel = _cos_rel(prm.grad, res["emu"][1][k].grad)
        cos2, _ = _cos_rel(prm.grad, res["emu2"][1][k].grad)
        cos_ee, _ = _cos_rel(res["emu"][1][k].grad, res["emu2"][1][k].grad)
        cos32, _ = _cos_rel(prm.grad, res["fp32"][1][k].grad)
        print(f"   {k:24s} HIP vs emulation: cos {cos:.6f} (rel {rel:.4f}), vs re-ordered emulation {cos2:.6f} | emulation vs re-ordered emulation: {cos_ee:.6f} | HIP vs fp32 oracle: {cos32:.5f}")
        checked += 1
        if k.startswith(("dconv_up1", "conv_last")) and not (cos >= 0.999 and rel <= 5e-2):
            bad.append(f"{k}: cos {cos:.6f} rel {rel:.4f} (no AdaIN backward upstream: must be tight)")
        if 1.0 - min(cos, cos2) > 2.0 * (1.0 - cos_ee) + 2e-3:
            bad.append(f"{k}: HIP-to-emulation distance {1 - min(cos, cos2):.4f} exceeds 2x the emulation-to-emulation distance {1 - cos_ee:.4f}")
    assert not bad, "; ".join(bad)
    assert checked == 36


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("fused", [True, False])
def test_reference_dropout_masks_injected(golden_dir, precision, fused):
    """tests/golden/cunet_train_b2_64.npz holds the REFERENCE's train-mode output together with the keep-masks its own
    nn.Dropout drew (cunet.py:61,68,75).  The HIP path takes those masks through `net.dropout_masks` (the kernels read the
    supplied keep bits instead of their counter RNG): output vs the reference within north_star's tolerance."""
    g = np.load(os.path.join(golden_dir, "cunet_train_b2_64.npz"))
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    s = size // 4
    masks = []
    for key, ch, hw in (("mask3", 512, s), ("mask2", 256, 2 * s), ("mask1", 128, 4 * s)):
        m = np.unpackbits(g[key])[:batch * ch * hw * hw].reshape(batch, ch, hw, hw)
        masks.append(torch.from_numpy(m.copy()))
    net = _make_g(nc, seed, precision).train()
    net.fused = fused
    net.dropout_masks = masks
    x, c = O.make_inputs(batch, size, nc, seed, True)
    xd, cd = x.to(DEV), c.to(DEV)
    with torch.no_grad():
        out = net(xd, cd)
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    print(f"reference train-mode fixture, {precision}, fused={fused}: max-abs {err:.3e}")
    assert err <= FWD_TOL[precision]
    # ... and the backward pass reads the same injected bits: gradients vs the oracle given the same masks
    out = net(xd, cd)
    torch.mean(torch.abs(out - xd)).backward()
    p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    O.bench_loss(O.cunet_forward(p, x, c, [m.float() for m in masks]), x).backward()
    for k in ("dconv_up3.0.weight", "dconv_down4.2.weight", "adain3.l1.weight", "dconv_down1.0.weight"):
        cos, _ = _cos_rel(dict(net.named_parameters())[k].grad, p[k].grad)
        assert cos >= (0.999 if precision == "fp32" else 0.93), f"{k}: {cos}"
    # masks are an override, not a mode: removing them restores the counter RNG
    net.dropout_masks = None
    net.dropout_seed = 3
    with torch.no_grad():
        assert not torch.equal(net(xd, cd), out.detach())


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: 512x512, B=16 (inference/inf_transfer_c.py:108-121)
# ---------------------------------------------------------------------------------------------------------------------------
def test_config4_512_b16():
    from wu.graph_infer import GraphedUNet
    nc, seed, B, S = 5, 2, 16, 512
    net = _make_g(nc, seed, "bf16").eval()
    g = torch.Generator().manual_seed(41)
    x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(DEV)
    eye = torch.eye(nc, device=DEV)
    c = eye[2].unsqueeze(0).expand(B, nc).contiguous()                       # inf_transfer_c.py:116: one-hot row tiled over the batch
    with torch.no_grad():
        y1 = net(x, c).clone()
        y2 = net(x, c)
    assert tuple(y1.shape) == (B, 3, S, S) and torch.equal(y1, y2)           # (1) eval forward is deterministic
    assert torch.isfinite(y1).all() and y1.abs().max().item() <= 1.0
    net32 = _make_g(nc, seed, "fp32").eval()
    with torch.no_grad():
        y32 = net32(x, c)
    err = (y32 - y1).abs().max().item()
    print(f"512x512 B=16: bf16 production kernels vs fp32 kernels max-abs {err:.3e}")
    assert err <= FWD_TOL["bf16"]                                            # (2) the 16 x 32 tile grid / DMA offset range at 512
    del net32, y32
    gr = GraphedUNet(net, batch=B, size=S)
    out = gr(x, c, copy_out=True)
    assert torch.equal(out, y1)                                              # (3) hipGraph replay == eager, bit for bit
    c2 = eye[4].unsqueeze(0).expand(B, nc).contiguous()
    with torch.no_grad():
        y_c2 = net(x, c2).clone()
    assert torch.equal(gr(x, c2, copy_out=True), y_c2) and not torch.equal(y_c2, y1)
    # batch independence at this size: image 5 alone (B=1 graph-free) == image 5 of the batch up to the batch-size-dependent
    # split of the instance statistics (see test_full_size_properties)
    with torch.no_grad():
        one = net(x[5:6].contiguous(), c[5:6].contiguous())
    assert (one - y1[5:6]).abs().max().item() <= 2e-2


def test_hipgraph_dropout_active_and_recapture():
    """The reference's inference loops run with Dropout ACTIVE (inference/inf_transfer_c.py:88-96 never calls .eval()).  A
    train-mode graph draws a NEW mask every replay through the device-resident seed counter, replay k == the eager module with
    dropout_seed = base + k; after a weight update the graph re-captures instead of replaying stale packed operands."""
    from wu.graph_infer import GraphedUNet
    nc, seed = 5, 8
    net = _make_g(nc, seed, "bf16").train()
    x, c = (t.to(DEV) for t in O.make_inputs(2, 64, nc, seed, True))
    gr = GraphedUNet(net, batch=2, size=64, base_seed=100)
    outs = [gr(x, c, copy_out=True) for _ in range(3)]
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    for k in range(3):
        net.dropout_seed = 100 + k
        with torch.no_grad():
            assert torch.equal(net(x, c), outs[k]), f"replay {k} != eager with dropout_seed {100 + k}"
    gr.set_seed_offset(1)
    assert torch.equal(gr(x, c, copy_out=True), outs[1])
    assert net.dropout_seed == 102 and net._seed_dev is None                 # capture leaves the module as it found it
    # weight update -> re-capture
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.05)
    gr.set_seed_offset(0)
    new = gr(x, c, copy_out=True)
    net.dropout_seed = 100
    with torch.no_grad():
        assert torch.equal(net(x, c), new) and not torch.equal(new, outs[0])
    # eval-mode graphs: no counter, identical replays, still follow weight updates
    net.eval()
    ge = GraphedUNet(net, batch=2, size=64)
    a = ge(x, c, copy_out=True)
    assert torch.equal(a, ge(x, c, copy_out=True)) and ge.seed_counter is None
    net.load_state_dict(O.make_cunet_params(nc, seed + 1))
    with torch.no_grad():
        assert torch.equal(ge(x, c, copy_out=True), net(x, c))


# ---------------------------------------------------------------------------------------------------------------------------
# advisor (high): packed MFMA operands of a spectral-norm conv must follow sigma and the optimizer
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_sndisc_packed_weights_follow_optimizer_steps(precision):
    import disc
    nc, seed = 5, 5
    d = _make_d(nc, seed, precision).train()
    opt = torch.optim.Adam(d.parameters(), lr=2e-2, betas=(0.0, 0.999))
    x, c = (t.to(DEV) for t in O.make_inputs(2, 64, nc, seed, True))
    fresh = disc.SNDisc(nc, precision=precision).to(DEV)
    for step in range(4):
        # a fresh module loaded from the current state must produce the very same train-mode forward (same power iteration)
        fresh.load_state_dict({k: v.clone() for k, v in d.state_dict().items()})
        fresh.train()
        with torch.no_grad():
            want = fresh(x, c)[0]
        opt.zero_grad()
        out = d(x, c)[0]
        assert torch.equal(out.detach(), want), f"step {step}: stale packed weights (max diff {(out.detach() - want).abs().max().item():.3e})"
        torch.mean(torch.relu(1.0 - out)).backward()
        opt.step()
    d.eval()
    fresh.load_state_dict(d.state_dict())
    fresh.eval()
    with torch.no_grad():
        a, b = d(x, c)[0], fresh(x, c)[0]
        assert torch.equal(a, b)
        assert torch.equal(d(x, c)[0], a)           # eval: sigma fixed, cached pack reused, same bits


# ---------------------------------------------------------------------------------------------------------------------------
# reducer on a real RCCL group (one rank; the collective path forced on)
# ---------------------------------------------------------------------------------------------------------------------------
def test_reducer_collective_path_on_rccl_world1():
    """`GradBucketReducer._launch` skips the collective when world == 1; with `world` forced to 2 on a one-rank RCCL group every
    bucket goes through a REAL async dist.all_reduce (AVG over one rank = identity).  Checks: each all-reduce issued from inside
    the fused backward is issued with the SIDE stream current (the process group's stream then waits for side >= main >= every
    producer, see wu.unet_graph.GradRouter), results are bit-identical to plain autograd over two steps, and a second backward
    without accumulate() raises."""
    import torch.distributed as dist
    from wu import ddp as D
    from wu import unet_graph as UG
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        nc = 5
        net = _make_g(nc, 4, "bf16").train()
        net.dropout_seed = 9
        x, c = (t.to(DEV) for t in O.make_inputs(2, 64, nc, 4, True))

        def run():
            torch.mean(torch.abs(net(x, c) - x)).backward()

        for p in net.parameters():
            p.grad = None
        run()
        ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        red = D.GradBucketReducer(D.ready_order(net), bucket_mb=2.0, ready_order=True).attach(net)
        red.world = 2                                  # test hook: take the collective path on the one-rank group
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **kw):
            calls.append(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
            return real(t, *a, **kw)
        D.dist.all_reduce = spy
        try:
            side = UG.prepare_side_stream(torch.device(DEV))
            for _ in range(2):
                red.zero_grad()
                calls.clear()
                run()
                in_backward = list(calls)
                red.finalize()
                assert len(in_backward) >= len(red.buckets) - 1 and len(calls) == len(red.buckets)
                assert all(s == side.cuda_stream for s in in_backward), "bucket all-reduce issued off the side stream"
                for k, p in net.named_parameters():
                    if k in ref:
                        assert torch.equal(p.grad, ref[k]), k
            red.zero_grad()
            run()
            with pytest.raises(RuntimeError, match="accumulate"):
                run()
            red.finalize()
            red.zero_grad()
            with red.accumulate():
                run()
                run()
                assert calls[-1:] != [] and red.launch_log == []
            red.finalize()
            torch.cuda.synchronize()
            for k in ("dconv_up1.2.weight", "dconv_down4.0.bias", "conv_last.weight"):
                assert torch.allclose(dict(net.named_parameters())[k].grad, 2 * ref[k], rtol=1e-5, atol=1e-8), k
        finally:
            D.dist.all_reduce = real
            net.grad_sink = None
            red.remove_hooks()
    finally:
        if created:
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------
# evaluation() as one batched pass
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["cls", "est"])
def test_batched_evaluation_matches_oracle_loop(mode):
    """WeatherTransferStep.evaluation (one (B*B)-image pass) against the oracle's restatement of the reference's B-pass loop
    (t_cls_train.py:314-367 / t_est_train.py:285-332), fp32, deterministic modes (G eval, D eval)."""
    from wu.train_step import WeatherTransferStep, StandInEstimator
    nc, seed, bs, size = 5, 6, 4, 64
    st = WeatherTransferStep(nc, mode=mode, precision="fp32", device=DEV, ddp=False, seed=1)
    st.inference.load_state_dict(O.make_cunet_params(nc, seed))
    st.discriminator.load_state_dict(O.make_sndisc_params(nc, seed))
    st.inference.eval()
    st.discriminator.eval()
    images, labels = O.make_inputs(bs, size, nc, seed, True)
    _, ref_labels = O.make_inputs(bs, size, nc, seed + 1, True)
    est_cpu = StandInEstimator(nc)
    est_cpu.load_state_dict({k: v.cpu() for k, v in st.estimator_.state_dict().items()})
    est_out = (lambda t: torch.softmax(est_cpu(t), 1)) if mode == "cls" else est_cpu
    est_eval = est_cpu if mode == "cls" else est_out            # self.estimator_ (:338) vs self.estimator (t_est_train.py:309)
    want, fakes = O.evaluation(O.make_cunet_params(nc, seed), O.make_sndisc_params(nc, seed), est_out, est_eval, images, labels, ref_labels)
    for max_images in (1024, 8):                                # one pass, and chunked
        got, fake = st.evaluation(images.to(DEV), labels.to(DEV), ref_labels.to(DEV), max_images=max_images)
        assert tuple(fake.shape) == (bs, bs, 3, size, size)
        assert (fake.cpu() - torch.stack(fakes)).abs().max().item() <= FWD_TOL["fp32"]
        for k, v in want.items():
            assert abs(got[k].item() - v) <= 2e-3 * max(1.0, abs(v)), (k, got[k].item(), v)
    # class-index labels (--one_hot, t_cls_train.py:327-329)
    idx = torch.arange(bs) % nc
    got_idx, _ = st.evaluation(images.to(DEV), idx.to(DEV), ref_labels.to(DEV))
    got_oh, _ = st.evaluation(images.to(DEV), torch.eye(nc)[idx].to(DEV), ref_labels.to(DEV))
    assert torch.equal(got_idx["d_loss"], got_oh["d_loss"])


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_sndisc_gradients_are_bitwise_reproducible(precision):
    """Round 1 left fp32 atomics in the weight gradients of SNDisc's image-layout convs (3->3 and 3->64 stride 2, every variant
    except the bf16 64-channel one); they now write per-workgroup partials into the caller's slab and are folded in workgroup
    order, like the other thin layers: every D gradient of the hinge loss, and the image gradient, repeat bit for bit."""
    nc, seed = 5, 5
    x, c = (t.to(DEV) for t in O.make_inputs(4, 96, nc, seed, True))
    runs = []
    for _ in range(3):
        d = _make_d(nc, seed, precision).train()
        xr = x.clone().requires_grad_(True)
        out = d(xr, c)[0]
        torch.mean(torch.relu(1.0 - out)).backward()
        runs.append(({k: p.grad.clone() for k, p in d.named_parameters()}, xr.grad.clone()))
    for g, gx in runs[1:]:
        assert torch.equal(gx, runs[0][1])
        for k in g:
            assert torch.equal(g[k], runs[0][0][k]), f"{k} differs between runs"


def test_fused_adam_updates_reach_the_convs():
    """Regression: torch.optim.Adam(fused=True) does not move ``param._version``; the packed MFMA operand images must follow the
    update anyway.  After two fused-Adam steps the net's output equals that of a FRESH net loaded with the same state-dict, and the
    packed image of a conv equals its current weight."""
    import cunet
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()
    x = (torch.rand(2, 3, 64, 64, device=dev) * 2 - 1)
    c = torch.eye(5, device=dev)[torch.arange(2) % 5]
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.0, 0.999), fused=True)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        net(x, c).abs().mean().backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        out = net(x, c)
    conv = net.dconv_down2[0]
    w = conv.weight.detach()
    ref = w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).to(torch.bfloat16)
    assert torch.equal(conv._packed.w_fwd, ref), "packed operand image is stale after a fused optimizer step"
    fresh = cunet.Conditional_UNet(5, precision="bf16").to(dev).eval()
    fresh.load_state_dict(net.state_dict())
    with torch.no_grad():
        assert torch.equal(fresh(x, c), out)


def test_gate_bit_producers_at_full_size():
    """The forward kernels that also write gate bits must produce the SAME activations as their plain forms at BASELINE size (B=32
    256x256): many iterations per wave / many tiles per workgroup, where the prefetch-and-counted-wait machinery is actually
    exercised (a register-prefetch bug of the 3->64 kernel passed every small-shape test), and the bits must equal (y > 0)."""
    from wu import kernels as K
    from wu.layout import empty_nhwc, precision_code
    dev = torch.device("cuda")
    code = precision_code("bf16")
    n = 32
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((n, 3, 256, 256), generator=g) * 2 - 1).to(dev)
    w = ((torch.rand((64, 3, 3, 3), generator=g) * 2 - 1) * 0.3).to(dev)
    b = (torch.rand((64,), generator=g) - 0.5).to(dev)
    y0 = empty_nhwc(n, 64, 256, 256, torch.bfloat16, dev)
    K.conv3x3_c3(x, w, b, y0, 1, 1, False, code)
    for _ in range(2):
        y1 = empty_nhwc(n, 64, 256, 256, torch.bfloat16, dev)
        bits = K.gate_bits_alloc(y1)
        K.conv3x3_c3_bits(x, w, b, y1, bits, 1, code)
        assert torch.equal(y0, y1)
        word = bits.view(n, 256, 256, 1, 2)
        pos = (y1 > 0)
        for hf in range(2):
            for k in range(4):
                for i in range(0, 8, 3):
                    got = (word[:, :, :, 0, hf] >> (8 * k + i)) & 1
                    assert torch.equal(got.bool(), pos[:, 16 * k + 8 * hf + i])
    for (ci, co, h) in [(64, 128, 128), (192, 64, 256), (256, 512, 32)]:
        xx = empty_nhwc(n, ci, h, h, torch.bfloat16, dev)
        xx.copy_((torch.rand((n, ci, h, h), generator=g) * 2 - 1).to(dev))
        ww = ((torch.rand((co, ci, 3, 3), generator=g) * 2 - 1) * 0.05).to(dev)
        wf, wd = K.pack_conv3x3(ww, code)
        bb = (torch.rand((co,), generator=g) - 0.5).to(dev)
        z0 = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
        K.conv3x3(xx, wf, bb, z0, 1, 1)
        z1 = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
        zb = K.gate_bits_alloc(z1)
        K.conv3x3_bits(xx, wf, bb, z1, 1, gate_bits_out=zb)
        assert torch.equal(z0, z1)
        # the data-gradient pass gated by those bits == gated by the tensor
        gy = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
        gy.copy_((torch.rand((n, co, h, h), generator=g) * 2 - 1).to(dev))
        wsq = ((torch.rand((co, co, 3, 3), generator=g) * 2 - 1) * 0.05).to(dev)
        _, wd2 = K.pack_conv3x3(wsq, code)
        d0 = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
        d1 = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
        K.conv3x3(gy, wd2, None, d0, 1, 0, egate=z1, egate_act=1)
        K.conv3x3_bits(gy, wd2, None, d1, 0, egate_bits=zb)
        assert torch.equal(d0, d1)


def test_three_step_trajectory_fused_vs_foreach_adam():
    """Multi-step parity: three Adam steps with the fused optimizer (no `_version` bump) and with the foreach optimizer (bumps it)
    must walk the same trajectory -- same losses, same parameters to fp32 rounding -- i.e. every forward sees the weights the last
    step wrote.  (Single-step oracle parity cannot see a stale operand cache; this can.)"""
    import cunet
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(21)
    x = (torch.rand((2, 3, 64, 64), generator=g) * 2 - 1).to(dev)
    c = torch.eye(5)[torch.arange(2) % 5].to(dev)

    def run(fused):
        torch.manual_seed(4)
        net = cunet.Conditional_UNet(5, precision="fp32").to(dev).train()
        net.dropout_seed = 9
        opt = torch.optim.Adam(net.parameters(), lr=2e-3, betas=(0.0, 0.999), fused=fused, foreach=None if fused else True)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = (net(x, c) - x).abs().mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}

    la, pa = run(True)
    lb, pb = run(False)
    assert la[0] == lb[0]
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(la, lb)), (la, lb)
    assert la[2] < la[0]
    # beta1 = 0: an element's step is lr * g / (|g| + eps) -- a SIGN function of g up to eps, ill-conditioned where g ~ 0: a single
    # element whose tiny gradient rounds to the other sign in one of the two implementations moves a full step the other way
    # (2 lr apart).  Step 1 sees identical gradients; the two implementations' last-bit differences then perturb the gradients of steps
    # 2 and 3, so the worst case is two opposite steps = 4 lr = 8e-3 (observed 1.2e-3 ... 5.0e-3 on single elements, depending on the
    # build); the bulk must agree to rounding (mean)
    # What the test guards against -- a forward that convolves with the weights of an EARLIER step -- moves every element by about lr per
    # step (mean difference ~1e-3, most elements above 1e-3): the bounds sit an order of magnitude below that signature and above the
    # sign-flip noise (a 512-element bias with two flipped elements has a mean difference of 2e-5)
    for k in pa:
        d = (pa[k].float() - pb[k].float()).abs()
        assert d.max().item() <= 8e-3 and d.mean().item() <= 1e-4 and (d > 1e-3).float().mean().item() <= 0.02, \
            f"{k}: fused and foreach Adam trajectories differ (max {d.max().item()}, mean {d.mean().item()}, share above 1e-3 {(d > 1e-3).float().mean().item()})"
