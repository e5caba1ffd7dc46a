"""Round-4 GPU tests: kernels and schedules added this round, each against the kernel / order it replaces or against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


# ---------------------------------------------------------------------------------------------------------------------------------
# AdaIN / bilinear x2 / dropout backward through the LDS ring (csrc/glue.hip: adain_upcat_bwd_tile_kernel, utils.py:41-51 + cunet.py:59-62)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [
    (2, 128, 24, 40, 64),      # two column tiles (the second one 8 columns wide), three strips, ragged right edge inside a tile
    (1, 64, 7, 33, 64),        # odd height, 33 columns: a one-column second tile; single strip
    (3, 256, 16, 32, 128),     # exactly one full column tile: the right halo pixel is outside the image
    (2, 64, 40, 70, 0),        # three column tiles, five strips, no skip channels behind the slice
    (1, 512, 2, 2, 256),       # smallest legal image
])
@pytest.mark.parametrize("p_drop", [0.3, 0.0])
def test_adain_upcat_bwd_lds_ring_vs_marching(shape, p_drop):
    """The LDS-ring kernel computes g' with the marching kernel's arithmetic term for term: the parked g' must be BIT-IDENTICAL; the
    per-(n, c) sums are grouped by other tiles (fp32 re-association) and dx follows from them within bf16 rounding."""
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc, nhwc_ld, stream_ptr
    n, c, h, w, cs = shape
    dev = _dev()
    bf = torch.bfloat16
    x = as_nhwc(_rand((n, c, h, w), 81, -1, 2).to(dev), _lib.BF16)
    g = as_nhwc(_rand((n, c + cs, 2 * h, 2 * w), 82).to(dev), _lib.BF16)
    ystd = _rand((n, c), 83, 0.5, 1.5).to(dev)
    ymean = _rand((n, c), 84).to(dev)
    stats = K.adain_stats(x, 1e-5)
    cat = empty_nhwc(n, c + cs, 2 * h, 2 * w, bf, dev)
    mb = K.adain_upcat(x, stats, ystd, ymean, cat, p_drop, 4711, True)
    res = {}
    try:
        for mode in (5, 1):
            _lib.call("wu_set_option", 8, mode)
            dx = torch.full((n, h, w, c), float("nan"), dtype=bf, device=dev).permute(0, 3, 1, 2)
            dstd, dmean = torch.empty((n, c), device=dev), torch.empty((n, c), device=dev)
            gtmp = torch.full((n, h, w, c), float("nan"), dtype=bf, device=dev)
            sums = torch.empty((n, c, 2 * (1 + K.MAX_SPLITS)), device=dev)
            _lib.call("wu_adain_upcat_bwd", g.data_ptr(), nhwc_ld(g), x.data_ptr(), nhwc_ld(x), stats.data_ptr(), ystd.data_ptr(),
                      dx.data_ptr(), nhwc_ld(dx), dstd.data_ptr(), dmean.data_ptr(), gtmp.data_ptr(), sums.data_ptr(),
                      n, h, w, c, float(p_drop), 4711, mb.data_ptr() if mb is not None else None, 1, _lib.BF16, stream_ptr())
            torch.cuda.synchronize()
            res[mode] = (gtmp.float().clone(), dstd.clone(), dmean.clone(), dx.float().clone())
    finally:
        _lib.call("wu_set_option", 8, 1)
    old, new = res[5], res[1]
    assert not torch.isnan(new[0]).any() and not torch.isnan(new[3]).any()
    assert torch.equal(old[0], new[0]), f"parked g' differs: {(old[0] - new[0]).abs().max().item()}"
    for a, b, name in ((old[1], new[1], "d y_std"), (old[2], new[2], "d y_mean")):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()) * (h * w) ** 0.5, name
    assert (old[3] - new[3]).abs().max().item() <= 1.6e-2 * max(1.0, old[3].abs().max().item())


# ---------------------------------------------------------------------------------------------------------------------------------
# the estimator's bf16 BACKWARD pinned block by block (wu/resnet.py; reference use site t_cls_train.py:247-250: G's whole weather-loss
# gradient flows through the frozen ResNet-101's data-gradient pass)
# ---------------------------------------------------------------------------------------------------------------------------------
def _cos(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))


def test_estimator_bf16_backward_block_by_block():
    """Every one of the 33 Bottlenecks of the bf16 production estimator against the oracle's bf16 EMULATION of that block
    (oracle/resnet_ref.py: BatchNorm folded, folded weights / stored activations / stored gradients rounded to bf16 at the points the HIP
    path stores them), each given the SAME stage input and the SAME gated upstream gradient the HIP stage had (wu.resnet.CAPTURE) -- so
    nothing is amplified across blocks and what is compared is the kernels' own arithmetic: stored block output within two bf16 ulps of
    scale, block-input gradient cosine >= 0.9999.  The end-to-end input-gradient cosine against the fp32 oracle (0.986, bounded by 33
    blocks of bf16 storage) is compared with the end-to-end EMULATION as well."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import resnet_ref as R
    from wu import resnet as RN
    dev, nc = _dev(), 5
    p = R.make_resnet101_params(nc, 3)
    est = RN.ResNet101Estimator(nc, precision="bf16")
    est.load_state_dict(p, strict=False)
    est = est.to(dev)
    x = _rand((2, 3, 64, 64), 5)
    wsum = _rand((2, nc), 6)
    xd = x.to(dev).requires_grad_(True)
    RN.CAPTURE = []
    try:
        out = est(xd)
        (out * wsum.to(dev)).sum().backward()
        torch.cuda.synchronize()
        cap = RN.CAPTURE
    finally:
        RN.CAPTURE = None
    names = [f"layer{li}.{b}" for li, (_, blocks, _) in enumerate(R.LAYERS, start=1) for b in range(blocks)]
    assert len(cap) == len(names) == 33
    worst_cos, worst_fwd = 1.0, 0.0
    for rec in cap:
        prefix = names[rec["block"]]
        xin, g_out = rec["xin"].float().cpu().contiguous(), rec["g_out"].float().cpu().contiguous()
        out_emu, g_emu = R.bottleneck_bf16_stage_grad(p, prefix, xin, g_out, rec["stride"])
        got_out, got_g = rec["out"].float().cpu(), rec["g_in"].float().cpu()
        fwd = (got_out - out_emu).abs().max().item() / max(1.0, out_emu.abs().max().item())
        cs = _cos(got_g, g_emu)
        worst_cos, worst_fwd = min(worst_cos, cs), max(worst_fwd, fwd)
        assert fwd <= 2 ** -7, f"{prefix}: stored output differs from the bf16 emulation by {fwd:.2e} of scale"
        assert cs >= 0.9999, f"{prefix}: block-input gradient cosine {cs:.6f} against the bf16 emulation"
    # end to end: against the fp32 oracle (precision-mode bound) and against the bf16 emulation of the whole network
    xe = x.clone().requires_grad_(True)
    oe = R.resnet101_forward_bf16(p, xe)
    (oe * wsum).sum().backward()
    xf = x.clone().requires_grad_(True)
    of = R.resnet101_forward(p, xf)
    (of * wsum).sum().backward()
    c_emu, c_f32 = _cos(xd.grad.cpu(), xe.grad), _cos(xd.grad.cpu(), xf.grad)
    c_emu_f32 = _cos(xe.grad, xf.grad)
    o_emu = (out.detach().cpu() - oe.detach()).abs().max().item() / max(1.0, oe.abs().max().item())
    print(f"estimator bf16: worst block cosine {worst_cos:.6f}, worst block output error {worst_fwd:.2e}; end-to-end input-gradient cosine "
          f"vs emulation {c_emu:.5f}, vs fp32 {c_f32:.5f} (emulation vs fp32 {c_emu_f32:.5f}); outputs vs emulation {o_emu:.2e}")
    assert o_emu <= 2e-2
    # the HIP path must be no further from the emulation than the emulation is from fp32 (plus slack): what separates it from fp32 is
    # the precision mode, not the kernels
    assert c_emu >= min(0.999, c_emu_f32 - 0.002), (c_emu, c_emu_f32)


# ---------------------------------------------------------------------------------------------------------------------------------
# SNDisc's conv trunk as one autograd node (wu/disc_graph.py; reference disc.py:27-32)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("case", ["d_update", "g_update_frozen", "feature_losses"])
def test_sndisc_fused_trunk_is_bitwise_the_per_layer_path(precision, case):
    """The single-node trunk launches the same kernels as the per-layer autograd path and applies the LeakyReLU gates in the
    data-gradient epilogues (same roundings): outputs, the four feature maps, all 20 parameter gradients, the input gradient and the
    power-iteration buffers are BIT-IDENTICAL -- for the discriminator update (parameter gradients), the generator update (D frozen:
    data gradient only, no weight-gradient kernels) and a loss that also reads the returned feature maps c1..c4."""
    import disc
    dev, nc = _dev(), 5
    x = _rand((3, 3, 64, 96), 11).to(dev)
    c = torch.softmax(_rand((3, nc), 12), dim=1).to(dev)
    torch.manual_seed(4)
    ref = disc.SNDisc(nc, precision=precision).to(dev).train()
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    res = {}
    saved = disc.FUSED_TRUNK
    try:
        for fused in (False, True):
            disc.FUSED_TRUNK = fused
            D = disc.SNDisc(nc, precision=precision).to(dev).train()
            D.load_state_dict(state)
            if case == "g_update_frozen":
                for q in D.parameters():
                    q.requires_grad_(False)
            xd = x.clone().requires_grad_(True)
            outs = D(xd, c)
            loss = torch.mean(torch.relu(1.0 - outs[0])) + torch.mean(torch.relu(1.0 + outs[0]))
            if case == "feature_losses":
                loss = loss + sum(o.float().abs().mean() for o in outs[1:]) * 0.1
            loss.backward()
            torch.cuda.synchronize()
            res[fused] = ([o.detach().float().clone() for o in outs], {k: q.grad.clone() for k, q in D.named_parameters() if q.grad is not None},
                          xd.grad.clone(), {k: v.clone() for k, v in D.state_dict().items() if k.endswith(("weight_u", "weight_v"))})
    finally:
        disc.FUSED_TRUNK = saved
    (o0, g0, dx0, b0), (o1, g1, dx1, b1) = res[False], res[True]
    for i, (a, b) in enumerate(zip(o0, o1)):
        assert torch.equal(a, b), f"output {i} differs"
    assert set(g0) == set(g1) and len(g0) == (0 if case == "g_update_frozen" else 20)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), f"gradient of {k} differs (max {(g0[k] - g1[k]).abs().max().item():.3e})"
    assert torch.isfinite(dx0).all() and dx0.abs().max().item() > 0
    assert torch.equal(dx0, dx1), f"input gradient differs (max {(dx0 - dx1).abs().max().item():.3e})"
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


# ---------------------------------------------------------------------------------------------------------------------------------
# stride-2 data gradient as four parity-class convs (csrc/conv3x3_mfma.hip SPARSE instances; nets.py:30-31 backward)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(2, 64, 128, 16, 32), (1, 128, 64, 11, 13), (3, 64, 64, 5, 7), (1, 256, 128, 34, 40), (2, 64, 64, 2, 2)])
@pytest.mark.parametrize("gated", [False, True])
def test_stride2_dgrad_parity_classes(precision, shape, gated):
    """dX of y = conv3x3(x, w, stride 2, pad 1) from a given dY: the four sparse-tap convs (one per parity class of the input site) against
    float64 autograd on the same (rounded) operands and against the zero-stuffing path they replace -- odd heights / widths (a last row /
    column without a partner), one-tile and many-tile images, with and without the LeakyReLU gate of the consumer in the epilogue."""
    import torch.nn.functional as F
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc, precision_code
    n, cin, cout, h, w = shape
    dev, code = _dev(), precision_code(precision)
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    rnd = lambda t: t.to(dt).float()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    wt = rnd(_rand((cout, cin, 3, 3), 21, -0.1, 0.1))
    gy = rnd(_rand((n, cout, ho, wo), 22))
    xg = rnd(_rand((n, cin, h, w), 23))                      # the tensor whose LeakyReLU gates dX (the previous block's output)
    x64 = torch.zeros((n, cin, h, w), dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, wt.double(), stride=2, padding=1).backward(gy.double())
    want = x64.grad.float()
    if gated:
        want = want * torch.where(xg > 0, torch.ones_like(xg), torch.full_like(xg, 0.2))
    _, wd = K.pack_conv3x3(wt.to(dev), code)
    gyd = as_nhwc(gy.to(dev), code)
    eg = as_nhwc(xg.to(dev), code) if gated else None
    res = {}
    PW3_DEFAULT = 2 + 8 + (128 << 5)
    # (option 14, option 15): the classes on the gathered-row LDS-DMA pipeline (bf16; forced for every size, on the whole chip and on 3 workgroups),
    # the classes on the register-staged tap-list kernel, the zero-stuffing path
    paths = {"gathered": (1, 16 + 2 + 8, 0), "gathered, 3 workgroups": (1, 16 + 2 + 8, 3), "tap lists": (1, 0, 0), "zero stuffing": (0, 0, 0)}
    try:
        for name, (o14, o15, cus) in paths.items():
            _lib.call("wu_set_option", 14, o14)
            _lib.call("wu_set_option", 15, o15)
            _lib.call("wu_set_option", 10, cus)
            dx = empty_nhwc(n, cin, h, w, dt, dev)
            dx.fill_(float("nan"))
            K.conv3x3_s2_dgrad(gyd, wd, dx, egate=eg, egate_act=K.ACT_LEAKY if gated else K.ACT_NONE)
            torch.cuda.synchronize()
            res[name] = dx.float().cpu()
    finally:
        _lib.call("wu_set_option", 14, 1)
        _lib.call("wu_set_option", 15, PW3_DEFAULT)
        _lib.call("wu_set_option", 10, 0)
    tol = (2e-4 if precision == "fp32" else 1.2e-2) * max(1.0, want.abs().max().item())
    for name in paths:
        assert not torch.isnan(res[name]).any(), f"{name}: sites left unwritten"
        err = (res[name] - want).abs().max().item()
        assert err <= tol, f"{name}: {err} vs tolerance {tol}"
    assert torch.equal(res["gathered"], res["gathered, 3 workgroups"])


# ---------------------------------------------------------------------------------------------------------------------------------
# two chained pointwise convs in one launch (csrc/resnet.hip conv1x1_chain_kernel; torchvision Bottleneck, classifier.py:106-112)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(37, 64, 256, 64), (300, 128, 512, 128), (64, 256, 1024, 256), (33, 256, 1024, 512), (5, 64, 256, 128), (100, 128, 512, 64)])
@pytest.mark.parametrize("mode", ["forward", "backward"])
def test_conv1x1_chain_is_bitwise_two_pointwise_launches(shape, mode):
    """wu_conv1x1_chain against two wu_conv1x1_fwd calls on the same operands -- forward form (bias + residual + ReLU, then bias + ReLU)
    and backward form (residual + ReLU gate, then ReLU gate) -- with a ragged row count (rows past M in the last 32-row tile), every K1
    the estimator uses and a C2 smaller than a wave's share: both outputs BIT-IDENTICAL, and within bf16 rounding of float64."""
    from wu import _lib, resnet as RN
    from wu.layout import empty_nhwc
    m, k1, c1, c2 = shape
    dev, bf = _dev(), torch.bfloat16
    n, h, w = 1, 1, m
    def t(c, seed, lo=-1.0, hi=1.0):
        return _rand((n, h, w, c), seed, lo, hi).to(bf).to(dev).permute(0, 3, 1, 2)
    x = t(k1, 1)
    wa = (_rand((c1, k1), 2, -1, 1) / k1 ** 0.5).to(bf).to(dev)
    wb = (_rand((c2, c1), 3, -1, 1) / c1 ** 0.5).to(bf).to(dev)
    res, g1, g2 = t(c1, 4), t(c1, 5), t(c2, 6)
    ba, bb = _rand((c1,), 7).to(dev), _rand((c2,), 8).to(dev)
    RELU, NONE = RN.RELU, RN.NONE
    y1a, y2a = empty_nhwc(n, c1, h, w, bf, dev), empty_nhwc(n, c2, h, w, bf, dev)
    y1b, y2b = empty_nhwc(n, c1, h, w, bf, dev), empty_nhwc(n, c2, h, w, bf, dev)
    for y in (y1a, y2a, y1b, y2b):
        y.fill_(float("nan"))
    assert _lib.load().wu_conv1x1_chain_supported(k1, c1, c2, _lib.BF16)
    if mode == "forward":
        RN.conv1x1_chain(x, RN.frag_pack(wa), ba, res, RELU, None, NONE, y1a, RN.frag_pack(wb), bb, RELU, None, NONE, y2a)
        RN.conv1x1(x, wa, ba, y1b, RELU, residual=res)
        RN.conv1x1(y1b, wb, bb, y2b, RELU)
        r1 = torch.relu(x.double().cpu().permute(0, 2, 3, 1) @ wa.double().cpu().t() + ba.double().cpu() + res.double().cpu().permute(0, 2, 3, 1))
        r2 = torch.relu(y1b.double().cpu().permute(0, 2, 3, 1) @ wb.double().cpu().t() + bb.double().cpu())
    else:
        RN.conv1x1_chain(x, RN.frag_pack(wa), None, res, NONE, g1, RELU, y1a, RN.frag_pack(wb), None, NONE, g2, RELU, y2a)
        RN.conv1x1(x, wa, None, y1b, NONE, residual=res, egate=g1, egate_act=RELU)
        RN.conv1x1(y1b, wb, None, y2b, NONE, egate=g2, egate_act=RELU)
        r1 = (x.double().cpu().permute(0, 2, 3, 1) @ wa.double().cpu().t() + res.double().cpu().permute(0, 2, 3, 1)) * (g1.double().cpu().permute(0, 2, 3, 1) > 0)
        r2 = (y1b.double().cpu().permute(0, 2, 3, 1) @ wb.double().cpu().t()) * (g2.double().cpu().permute(0, 2, 3, 1) > 0)
    torch.cuda.synchronize()
    assert not torch.isnan(y1a.float()).any() and not torch.isnan(y2a.float()).any()
    assert torch.equal(y1a, y1b), f"y1 differs from the stand-alone launch: {(y1a.float() - y1b.float()).abs().max().item():.3e}"
    assert torch.equal(y2a, y2b), f"y2 differs from the stand-alone launch: {(y2a.float() - y2b.float()).abs().max().item():.3e}"
    for got, want in ((y1a, r1), (y2a, r2)):
        err = (got.double().cpu().permute(0, 2, 3, 1) - want).abs().max().item()
        assert err <= 1.2e-2 * max(1.0, want.abs().max().item()), err


def test_estimator_chained_pointwise_is_bitwise_the_unchained_network():
    """The whole bf16 ResNet-101 (33 Bottlenecks, 64x64 input) with and without the chained launches: outputs and the input gradient
    bit-identical; under no_grad as well (the reference's three no-grad estimator calls, t_cls_train.py:237,297,424)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import resnet_ref as R
    from wu import resnet as RN
    dev, nc = _dev(), 5
    p = R.make_resnet101_params(nc, 4)
    x = _rand((3, 3, 64, 96), 9)
    wsum = _rand((3, nc), 10).to(dev)
    res = {}
    saved = (RN.CHAIN, RN.CHAIN_MIN_K1)
    try:
        for chain in (False, True):
            RN.CHAIN, RN.CHAIN_MIN_K1 = chain, 64           # every pair the kernel supports, not only the ones the default picks
            est = RN.ResNet101Estimator(nc, precision="bf16")
            est.load_state_dict(p, strict=False)
            est = est.to(dev)
            xd = x.to(dev).requires_grad_(True)
            out = est(xd)
            (out * wsum).sum().backward()
            with torch.no_grad():
                out_ng = est(x.to(dev))
            torch.cuda.synchronize()
            res[chain] = (out.detach().clone(), xd.grad.clone(), out_ng.clone())
    finally:
        RN.CHAIN, RN.CHAIN_MIN_K1 = saved
    for a, b, name in zip(res[False], res[True], ("outputs", "input gradient", "no-grad outputs")):
        assert torch.isfinite(a).all() and a.abs().max().item() > 0
        assert torch.equal(a, b), f"{name}: chained != unchained (max {(a - b).abs().max().item():.3e})"


# ---------------------------------------------------------------------------------------------------------------------------------
# gate + arg-max bits from the fused pool epilogue, max-pool backward from bits (conv3x3_mfma_v2 GATED = 4, glue.hip; cunet.py:46,49,52)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 64), (1, 128, 128, 48, 96), (2, 256, 256, 16, 32), (1, 64, 64, 34, 38)])
@pytest.mark.parametrize("sliced", [False, True])
def test_pool_epilogue_bits_and_maxpool_backward_from_bits(shape, sliced):
    """The forward + pool + bits conv instance: y and the pooled tensor bit-identical to the plain fused-pool conv; the two bit planes equal
    to (y > 0) and to torch's max_pool2d arg-max (first maximum in scan order, ties included: ReLU outputs are full of equal zeros); and the
    bits-based backward bit-identical to the tensor-based one (with the skip gradient, incl. an output that is a channel slice of a
    concat buffer, 8-wave and 4-wave conv instances, ragged tiles)."""
    import torch.nn.functional as F
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc
    n, cin, cout, h, w = shape
    dev, bf = _dev(), torch.bfloat16
    x = as_nhwc(torch.relu(_rand((n, cin, h, w), 31)).to(dev), _lib.BF16)
    wt = _rand((cout, cin, 3, 3), 32, -0.1, 0.1)
    bias = _rand((cout,), 33, -0.3, 0.1).to(dev)
    wf, _ = K.pack_conv3x3(wt.to(dev), _lib.BF16)
    extra = 64 if sliced else 0
    buf_a, buf_b = empty_nhwc(n, extra + cout, h, w, bf, dev), empty_nhwc(n, extra + cout, h, w, bf, dev)
    ya, yb = buf_a[:, extra:], buf_b[:, extra:]
    if not K.gate_bits_supported(x, ya):
        pytest.skip("shape outside the LDS-DMA conv")
    pa, pb = empty_nhwc(n, cout, h // 2, w // 2, bf, dev), empty_nhwc(n, cout, h // 2, w // 2, bf, dev)
    gb, sb = K.gate_bits_alloc(ya), K.gate_bits_alloc(ya)
    gb.fill_(0x55555555); sb.fill_(0x55555555)
    K.conv3x3_relu_pool(x, wf, bias, ya, pa)
    K.conv3x3_relu_pool_bits(x, wf, bias, yb, pb, gb, sb)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb) and torch.equal(pa, pb)
    # decode the bit planes: word (pixel, ct, hf), bit 8k + i  <->  channel 64 ct + 16 k + 8 hf + i
    def decode(bits):
        wv = bits.view(n, h, w, cout // 64, 2).to(torch.int64) & 0xffffffff
        out = torch.zeros((n, h, w, cout), dtype=torch.bool, device=dev)
        for k in range(4):
            for i in range(8):
                b = ((wv >> (8 * k + i)) & 1).bool()                     # (n, h, w, ct, hf)
                for hf in range(2):
                    out[..., torch.arange(cout // 64, device=dev) * 64 + 16 * k + 8 * hf + i] = b[..., hf]
        return out.permute(0, 3, 1, 2)
    yf = ya.float()
    assert torch.equal(decode(gb), yf > 0), "gate bits"
    _, idx = F.max_pool2d(yf, 2, return_indices=True)                    # flat index into (h, w) of each window's first maximum
    want_sel = torch.zeros((n, cout, h * w), dtype=torch.bool, device=dev)
    want_sel.scatter_(2, idx.reshape(n, cout, -1), True)
    assert torch.equal(decode(sb), want_sel.view(n, cout, h, w)), "arg-max bits"
    # backward: bits vs tensor
    gy = as_nhwc(_rand((n, cout, h // 2, w // 2), 34).to(dev), _lib.BF16)
    gs = as_nhwc(_rand((n, cout, h, w), 35).to(dev), _lib.BF16)
    for dskip in (gs, None):
        d1, d2 = empty_nhwc(n, cout, h, w, bf, dev), empty_nhwc(n, cout, h, w, bf, dev)
        K.maxpool2_bwd(ya, gy, d1, dskip=dskip, gate_act=K.ACT_RELU)
        K.maxpool2_bwd_bits(gb, sb, gy, d2, dskip=dskip)
        torch.cuda.synchronize()
        assert torch.equal(d1, d2), f"max-pool backward from bits differs (skip gradient: {dskip is not None})"


# (N, Cin, Cout, H, W): the estimator's stride-1 pointwise shapes at a small batch, a ragged row count, a single K step and a single tile
PW3_SHAPES = [(2, 256, 1024, 16, 16), (2, 1024, 256, 16, 16), (3, 128, 512, 10, 10), (1, 512, 2048, 8, 8), (1, 2048, 512, 8, 8),
              (2, 64, 256, 24, 20), (2, 512, 128, 9, 7), (1, 192, 128, 5, 5), (2, 256, 64, 24, 20), (1, 128, 192, 9, 9), (3, 64, 64, 16, 16)]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [16 + 2, 16 + 3, 16 + 4, 16 + 8 + 2, 16 + 8 + 3, 16 + 8 + 4])
@pytest.mark.parametrize("shape", PW3_SHAPES)
def test_pointwise_persistent_dma_kernel_is_bit_identical(shape, depth):
    """conv1x1_pw3_kernel (round 4: 128 x 128 tiles, LDS-DMA ring of `depth & 7` stages, four or -- bit 3 -- eight waves, 64-cout tiles allowed -- bit 4 --, persistent workgroups, counted waits) against
    conv1x1_mfma_kernel on the same operands, bitwise: plain; bias + residual + ReLU (the forward of a Bottleneck's conv3); gate only (data
    gradient of conv3); residual + gate (data gradient of conv1 plus the identity path).  Outputs and residuals are channel slices of wider
    buffers (ld > C), the grid is held at 5 workgroups so that every workgroup walks several tiles and the ring wraps across tile borders."""
    from wu import _lib, resnet as RN
    from wu.layout import as_nhwc, empty_nhwc
    dev = _dev()
    bf = torch.bfloat16
    n, cin, cout, h, w = shape
    x = as_nhwc(_rand((n, cin, h, w), 41).to(dev), _lib.BF16)
    wt = _rand((cout, cin), 42, -0.1, 0.1).to(dev).to(bf).contiguous()
    b = _rand((cout,), 43).to(dev)
    res_buf = as_nhwc(_rand((n, cout + 64, h, w), 44).to(dev), _lib.BF16)
    gate_buf = as_nhwc(_rand((n, cout + 128, h, w), 45).to(dev), _lib.BF16)
    res, gate = res_buf[:, 64:], gate_buf[:, :cout]
    OPT_PW3, OPT_GRID = 15, 10
    cases = [dict(), dict(act=1, residual=res), dict(egate=gate, egate_act=1), dict(residual=res, egate=gate, egate_act=1), dict(act=1)]
    try:
        for ci, kw in enumerate(cases):
            bias = None if ci in (2, 3) else b
            outs = []
            for opt, cus in ((0, 0), (depth, 0), (depth, 5)):
                _lib.call("wu_set_option", OPT_PW3, opt)
                _lib.call("wu_set_option", OPT_GRID, cus)
                ybuf = empty_nhwc(n, cout + 64, h, w, bf, dev)
                ybuf.fill_(7.0)
                RN.conv1x1(x, wt, bias, ybuf[:, :cout], **kw)
                torch.cuda.synchronize()
                outs.append(ybuf)
            assert torch.equal(outs[0], outs[1]), f"case {ci}: persistent kernel differs (or wrote outside its channels)"
            assert torch.equal(outs[0], outs[2]), f"case {ci}: persistent kernel on 5 workgroups differs"
            assert bool((outs[0][:, cout:] == 7.0).all())
    finally:
        _lib.call("wu_set_option", OPT_PW3, 2 + 8 + (128 << 5))        # the library's default (wu_prof.hip)
        _lib.call("wu_set_option", OPT_GRID, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32), (3, 128, 256, 17, 23), (1, 256, 512, 9, 16), (2, 128, 128, 45, 70), (1, 512, 512, 16, 16),
                                   (2, 64, 64, 20, 28), (1, 128, 192, 11, 7)])
def test_stride2_conv_on_the_gathered_row_pipeline(shape):
    """wu_conv3x3_fwd, stride 2, bf16: the gathered-row form of the persistent LDS-DMA GEMM (round 4, conv1x1_pw3_kernel<.., CONV>) against
    F.conv2d on the bf16-rounded operands (tolerance of the bf16 conv tests) and against the register-staged kernel it replaces (option 15 = 0;
    the K order differs -- (64-channel chunk, tap) against (32-channel chunk, tap) -- so the two may differ in the last bf16 digit): bias +
    LeakyReLU (SNDisc), bias + ReLU with an output gate, on 5 workgroups (several tiles per workgroup) and on the whole chip; odd image sizes."""
    import torch.nn.functional as F
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc
    dev = _dev()
    bf = torch.bfloat16
    n, cin, cout, h, w = shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    x32 = _rand((n, cin, h, w), 51).to(bf).float()
    wt = _rand((cout, cin, 3, 3), 52, -0.05, 0.05)
    b = _rand((cout,), 53, -0.2, 0.2)
    x = as_nhwc(x32.to(dev), _lib.BF16)
    wf, _ = K.pack_conv3x3(wt.to(dev), _lib.BF16)
    gate = as_nhwc(_rand((n, cout, ho, wo), 54).to(dev), _lib.BF16)
    ref = F.conv2d(x32, wt.to(bf).float(), b, stride=2, padding=1)
    OPT_PW3, OPT_GRID, DEFAULT = 15, 10, 2 + 8 + (128 << 5)
    try:
        for act, eg in ((K.ACT_LEAKY, None), (K.ACT_RELU, gate)):
            want = F.leaky_relu(ref, 0.2) if act == K.ACT_LEAKY else F.relu(ref) * (gate.float().cpu() > 0).float()
            outs = []
            for opt, cus in ((0, 0), (16 + 2 + 8, 0), (16 + 2 + 8, 5)):
                _lib.call("wu_set_option", OPT_PW3, opt)
                _lib.call("wu_set_option", OPT_GRID, cus)
                ybuf = empty_nhwc(n, cout + 64, ho, wo, bf, dev)
                ybuf.fill_(7.0)
                K.conv3x3(x, wf, b.to(dev), ybuf[:, :cout], 2, act, egate=eg, egate_act=K.ACT_RELU if eg is not None else K.ACT_NONE)
                torch.cuda.synchronize()
                assert bool((ybuf[:, cout:] == 7.0).all())
                outs.append(ybuf[:, :cout].float().cpu())
            scale = max(1.0, want.abs().max().item())
            for o in outs:
                assert (o - want).abs().max().item() <= 1.5e-2 * scale
            assert torch.equal(outs[1], outs[2]), "5 workgroups against the whole chip"
            assert (outs[0] - outs[1]).abs().max().item() <= 8e-3 * scale, "gathered-row form against the register-staged kernel"
    finally:
        _lib.call("wu_set_option", OPT_PW3, DEFAULT)
        _lib.call("wu_set_option", OPT_GRID, 0)


# ---------------------------------------------------------------------------------------------------------------------------------
# last decoder conv + 1x1 head + tanh in ONE launch (csrc/conv3x3_mfma_v2.hip GATED = 5; cunet.py:78-82)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [
    (2, 64, 64, 64),       # the network's own form (64 -> 64, weights resident in LDS), interior tiles only
    (3, 64, 40, 48),       # ragged: a half-empty bottom tile row and a half-empty right tile column
    (1, 192, 32, 96),      # six K chunks per tile (weights re-fetched), three column tiles
    (33, 64, 32, 32),      # more tiles than one pass of a small grid: the deferred stores of several tiles per workgroup
])
def test_conv_relu_head_fused_vs_two_launches(shape):
    """y must be BIT-IDENTICAL to the plain conv's (same K loop, same epilogue); the head agrees with the stand-alone fp32 head kernel on the
    same stored bf16 y to fp32 rounding (its weights enter the MFMA as bf16 high part + bf16 residual, products exact in fp32; only the
    summation order differs), and with a float64 evaluation of tanh(W y + b); without y the image is the same bit for bit."""
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc
    n, cin, h, w = shape
    dev = _dev()
    bf = torch.bfloat16
    x = as_nhwc(_rand((n, cin, h, w), 11, -1, 2).to(dev), _lib.BF16)
    wt = (_rand((64, cin, 3, 3), 12) * (2.0 / (9 * cin)) ** 0.5 * 1.7).to(dev)
    bias = _rand((64,), 13, -0.2, 0.2).to(dev)
    hw_ = (_rand((3, 64), 14) * 0.35).to(dev)
    hb = _rand((3,), 15, -0.3, 0.3).to(dev)
    assert K.conv3x3_head_supported(x)
    wf, _ = K.pack_conv3x3(wt, _lib.BF16)
    y_ref = K.conv3x3(x, wf, bias, empty_nhwc(n, 64, h, w, bf, dev), 1, K.ACT_RELU)
    out_ref = K.conv1x1_tanh(y_ref, hw_, hb, torch.empty((n, 3, h, w), device=dev))
    y = torch.full((n, h, w, 64), float("nan"), dtype=bf, device=dev).permute(0, 3, 1, 2)
    out = torch.full((n, 3, h, w), float("nan"), device=dev)
    K.conv3x3_relu_head(x, wf, bias, y, hw_, hb, out)
    out2 = torch.full((n, 3, h, w), float("nan"), device=dev)
    K.conv3x3_relu_head(x, wf, bias, None, hw_, hb, out2)
    torch.cuda.synchronize()
    assert torch.equal(y.float(), y_ref.float())
    assert not torch.isnan(out).any()
    z = torch.einsum("kc,nchw->nkhw", hw_.double().cpu(), y_ref.double().cpu()) + hb.double().cpu().view(1, 3, 1, 1)
    want = torch.tanh(z)
    assert (out.double().cpu() - want).abs().max().item() < 2e-6
    assert (out - out_ref).abs().max().item() < 2e-6
    assert torch.equal(out, out2)


def test_unet_forward_with_and_without_the_fused_head():
    """The whole generator, training mode (same dropout seed) and eval mode: outputs within fp32 rounding of the two-launch path and every
    gradient within fp32 rounding too (the backward is the same kernels on a bit-identical y and an image that differs by ~1e-7)."""
    import cunet
    from wu import unet_graph as UG
    dev = _dev()
    torch.manual_seed(3)
    net = cunet.Conditional_UNet(5, precision="bf16").to(dev)
    x = _rand((2, 3, 64, 96), 21).to(dev)
    c = torch.eye(5)[[1, 3]].to(dev)
    res = {}
    try:
        for fused in (True, False):
            UG.HEAD_FUSED = fused
            net.train()
            net.dropout_seed = 5
            net.zero_grad(set_to_none=True)
            xg = x.clone().requires_grad_(True)
            out = net(xg, c)
            (out - x).abs().mean().backward()
            grads = [p.grad.clone() for p in net.parameters() if p.grad is not None] + [xg.grad.clone()]
            assert len(grads) >= 37
            net.eval()
            with torch.no_grad():
                oe = net(x, c)
            torch.cuda.synchronize()
            res[fused] = (out.detach().clone(), grads, oe.clone())
    finally:
        UG.HEAD_FUSED = True
    a, b = res[True], res[False]
    assert (a[2] - b[2]).abs().max().item() < 2e-6
    assert (a[0] - b[0]).abs().max().item() < 2e-6
    for ga, gb in zip(a[1], b[1]):
        # the L1 loss's sign(out - x) can flip where out == x to the last bit: allow a few elements' worth of 1 / numel
        assert (ga - gb).abs().max().item() <= 2e-3 * max(1e-6, gb.abs().max().item())


# ---------------------------------------------------------------------------------------------------------------------------------
# 3x3 convs on images at most 16 pixels wide (csrc/conv3x3_small.hip: the estimator's layer3 / layer4, classifier.py:106)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    (5, 256, 256, 16, 16, False),     # layer3's shape: one image = two 8-row tiles
    (5, 128, 64, 8, 8, False),        # layer4's geometry: two 8x8 images stacked per tile, an odd batch (a half-empty last tile)
    (3, 64, 128, 13, 9, True),        # ragged, gated (data-gradient form): tile rows and columns past the image
    (7, 96, 64, 5, 3, False),         # 4-wide tiles, six 5-row images per 32-row tile and two unused rows
    (9, 64, 64, 4, 4, True),          # eight 4x4 images per tile
    (2, 128, 128, 24, 16, True),      # three row tiles per image
    (1, 32, 64, 1, 1, False),         # a single pixel
    (40, 64, 64, 2, 2, True),         # 2x2 images: thirteen per tile (the stacked halo image is capped at 320 pixels)
])
def test_small_image_conv_vs_generic_and_fp32(case):
    """Same products, another summation grouping (the K range split between wave pairs): within two bf16 ulps of the generic template's
    output and as close to an fp32 conv of the same bf16 operands as that one is."""
    import torch.nn.functional as F
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc
    n, ci, co, h, w, gated = case
    dev = _dev()
    x = as_nhwc(_rand((n, ci, h, w), 31, -0.3, 0.7).to(dev), _lib.BF16)
    wt = (_rand((co, ci, 3, 3), 32) * (6.0 / (9 * ci)) ** 0.5).to(dev)
    wf, _ = K.pack_conv3x3(wt, _lib.BF16)
    b = _rand((co,), 33, -0.5, 0.5).to(dev)
    eg = as_nhwc(_rand((n, co, h, w), 34).to(dev), _lib.BF16) if gated else None
    res = {}
    try:
        for opt in (1, 0):
            _lib.call("wu_set_option", 3, 2 if opt else 0)
            y = torch.full((n, h, w, co), float("nan"), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
            K.conv3x3(x, wf, None if gated else b, y, 1, K.ACT_NONE if gated else K.ACT_RELU, egate=eg, egate_act=K.ACT_RELU if gated else K.ACT_NONE)
            torch.cuda.synchronize()
            res[opt] = y.float().cpu()
    finally:
        _lib.call("wu_set_option", 3, 1)
    assert not torch.isnan(res[1]).any()
    # chunk-major weights (wu_conv3x3_small_fwd, the estimator's form): the same kernel reading the same numbers from another address
    yc = torch.full((n, h, w, co), float("nan"), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
    K.conv3x3_small(x, K.chunk_major(wf), None if gated else b, yc, K.ACT_NONE if gated else K.ACT_RELU, egate=eg, egate_act=K.ACT_RELU if gated else K.ACT_NONE)
    torch.cuda.synchronize()
    assert torch.equal(yc.float().cpu(), res[1])
    ref = F.conv2d(x.float().cpu(), wt.to(torch.bfloat16).float().cpu(), None if gated else b.cpu(), padding=1)
    ref = ref * (eg.float().cpu() > 0) if gated else torch.relu(ref)
    scale = max(1.0, ref.abs().max().item())
    assert (res[1] - res[0]).abs().max().item() <= 2 * scale * 2.0 ** -8
    assert (res[1] - ref).abs().max().item() <= scale * 2.0 ** -8 + 1e-6


def test_dropout_masks_equal_the_numpy_restatement_of_the_hash():
    """wu_dropout_mask (and with it every kernel that draws keep bits: they share wu_rand4) against tests/_dropout_hash.py, bit for bit --
    the statistics pinned on the CPU (tests/test_host_cpu.py::test_dropout_hash_statistics) are the kernels' statistics."""
    import numpy as np
    from _dropout_hash import keep_mask_nchw
    from wu import functional as WF
    dev = _dev()
    for (n, c, h, w, p, seed) in ((2, 128, 12, 20, 0.3, 133), (1, 512, 8, 8, 0.3, (99 * 4 + 3) & 0x7FFFFFFFFFFFFFFF), (3, 64, 6, 10, 0.5, (1 << 40) + 7)):
        got = WF.dropout_mask(n, c, h, w, p, seed, dev).cpu().numpy()
        assert np.array_equal(got, keep_mask_nchw(n, c, h, w, p, seed)), (n, c, h, w, p, seed)


def test_graphed_estimator_pass_is_bitwise_the_eager_pass():
    """The frozen estimator's no-grad pass (t_cls_train.py:424,297,237) replayed from a hipGraph (wu.resnet.GraphedEstimatorPass, what
    WeatherTransferStep.step() runs): raw outputs bit-equal to the eager launches for two different batches through ONE capture (the captured
    kernels read the static input, nothing of the warm-up data survives), the static output is not aliased by what the call returns, a wrong
    batch shape raises, and a moved estimator state (load_state_dict) is reported stale."""
    from wu.resnet import GraphedEstimatorPass, ResNet101Estimator
    DEV = _dev()
    torch.manual_seed(3)
    est = ResNet101Estimator(num_classes=5, precision="bf16").to(DEV)
    a, b = torch.randn(3, 3, 64, 64, device=DEV), torch.randn(3, 3, 64, 64, device=DEV)
    g = GraphedEstimatorPass(est, (6, 3, 64, 64))
    with torch.no_grad():
        e1 = est(torch.cat([a, b]))
        e2 = est(torch.cat([b, a]))
    r1 = g((a, b))
    r2 = g((b, a))
    torch.cuda.synchronize()
    assert torch.isfinite(e1).all() and (e1 != e2).any()
    assert torch.equal(r1, e1) and torch.equal(r2, e2), "graph replay differs from the eager pass"
    assert r1.data_ptr() != g.out.data_ptr()
    with pytest.raises(ValueError):
        g((a,))
    with pytest.raises(ValueError):
        g((a, torch.randn(3, 3, 32, 64, device=DEV)))
    assert not g.stale()
    sd = {k: v.clone() for k, v in est.state_dict().items()}
    sd["fc.bias"] = sd["fc.bias"] + 1
    est.load_state_dict(sd)
    assert g.stale()


def test_graphed_estimator_grad_pass_is_bitwise_the_eager_node():
    """The estimator's DIFFERENTIATED forward (t_cls_train.py:247-250) replayed from a hipGraph with its activations in the graph's static
    buffers (wu.resnet.GraphedEstimatorGradPass) and the eager backward on them: raw outputs and the input gradient bit-equal to the eager
    autograd node, for two different inputs through one capture; a second replay before the first one's backward makes that backward raise
    instead of differentiating through overwritten activations."""
    from wu.resnet import GraphedEstimatorGradPass, ResNet101Estimator
    DEV = _dev()
    torch.manual_seed(4)
    est = ResNet101Estimator(num_classes=5, precision="bf16").to(DEV)
    g = GraphedEstimatorGradPass(est, (4, 3, 64, 96))
    tgt = torch.randn(4, 5, device=DEV)
    for seed in (1, 2):
        x0 = _rand((4, 3, 64, 96), seed).to(DEV)
        xe = x0.clone().requires_grad_(True)
        ye = est(xe)
        ((ye - tgt) ** 2).sum().backward()
        xg = x0.clone().requires_grad_(True)
        yg = g(xg)
        ((yg - tgt) ** 2).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(yg, ye), "graphed forward differs from the eager node"
        assert xg.grad is not None and xg.grad.abs().max().item() > 0
        assert torch.equal(xg.grad, xe.grad), "input gradient through the graphed pass differs from the eager node"
    xa = _rand((4, 3, 64, 96), 7).to(DEV).requires_grad_(True)
    ya = g(xa)
    g(_rand((4, 3, 64, 96), 8).to(DEV).requires_grad_(True))
    with pytest.raises(RuntimeError, match="replayed again"):
        ya.sum().backward()


@pytest.mark.parametrize("n,nc", [(32, 5), (3, 7), (9, 32)])
def test_adain_style_multi_is_bitwise_the_per_layer_calls(n, nc):
    """The three AdaIN style MLPs of a U-Net pass (utils.py:41-48 at cunet.py:59,66,73) in one launch per direction
    (wu_adain_style_{fwd,bwd}_multi) against three single calls: (y_std, y_mean) and the gradients into l1.weight / l1.bias bit-equal,
    channel counts that differ per level (the level's row range ends inside a wave), N not a multiple of the 8 sample lanes; then the
    whole network's forward and all 36 gradients with the batched call on and off."""
    from wu import functional as WF, unet_graph as UG
    DEV = _dev()
    chans = (512, 256, 100)
    y = _rand((n, nc), 1).to(DEV)
    layers, single = [], []
    for i, c in enumerate(chans):
        w = _rand((4 * c, nc), 10 + i).to(DEV).requires_grad_(True)
        b = _rand((4 * c,), 20 + i).to(DEV).requires_grad_(True)
        layers.append((w, b, 1e-5 * (i + 1)))
        single.append((w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)))
    outs = WF.adain_style_multi(y, layers)
    loss = 0
    for i, (sd, mn) in enumerate(outs):
        loss = loss + (sd * _rand(sd.shape, 30 + i).to(DEV)).sum() + (mn * _rand(mn.shape, 40 + i).to(DEV)).sum()
    loss.backward()
    for i, ((w1, b1), (w, b, eps)) in enumerate(zip(single, layers)):
        sd, mn = WF.adain_style(y, w1, b1, eps)
        ((sd * _rand(sd.shape, 30 + i).to(DEV)).sum() + (mn * _rand(mn.shape, 40 + i).to(DEV)).sum()).backward()
        assert torch.equal(sd, outs[i][0]) and torch.equal(mn, outs[i][1]), f"level {i}: forward"
        assert torch.isfinite(w.grad).all() and w.grad.abs().max().item() > 0
        assert torch.equal(w.grad, w1.grad) and torch.equal(b.grad, b1.grad), f"level {i}: gradients"
    if (n, nc) != (32, 5):
        return
    import cunet
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision="bf16").to(DEV).train()
    net.dropout_seed = 3
    x = _rand((2, 3, 64, 64), 5).to(DEV)
    c = torch.eye(5, device=DEV)[[1, 3]]
    res = []
    saved = UG.STYLE_BATCHED
    try:
        for flag in (True, False):
            UG.STYLE_BATCHED = flag
            net.dropout_seed = 3
            net.zero_grad(set_to_none=True)
            out = net(x, c)
            torch.mean(torch.abs(out - x)).backward()
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    finally:
        UG.STYLE_BATCHED = saved
    assert torch.equal(res[0][0], res[1][0])
    assert len(res[0][1]) == 36 and set(res[0][1]) == set(res[1][1])      # the reference's 36 trained tensors (the unused embeddings get none)
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Dropout keep decisions (cunet.py:61,68,75): stored keep bytes (default) against drawn again by the backward (wu.unet_graph.KEEP_BITS_STORED,
# WU_KEEP_BITS=0; measured not faster: profiles/r04_keep_bits_ab.txt).  Same masks: the forward is bit-identical; the backward kernels (LDS ring
# reading bytes / marching kernel re-hashing, themselves compared bit for bit in test_gpu_kernels.py::test_adain_upcat_bwd_mask_bits_vs_rehash)
# partition the per-(n, c) sums differently, so the gradients agree to bf16 rounding.
# ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_dropout_keep_bytes_stored_vs_redrawn_by_the_backward():
    import cunet
    from wu import unet_graph as UG
    DEV = _dev()
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision="bf16").to(DEV).train()
    x = _rand((4, 3, 64, 64), 5).to(DEV)
    c = torch.eye(5, device=DEV)[[1, 3, 0, 4]]
    res = []
    saved = UG.KEEP_BITS_STORED
    try:
        for flag in (True, False):
            UG.KEEP_BITS_STORED = flag
            net.dropout_seed = 11
            net.zero_grad(set_to_none=True)
            out = net(x, c)
            torch.mean(torch.abs(out - x)).backward()
            res.append((out.detach().clone(), {k: p.grad.double().clone() for k, p in net.named_parameters() if p.grad is not None}))
    finally:
        UG.KEEP_BITS_STORED = saved
    assert torch.equal(res[0][0], res[1][0])
    assert len(res[0][1]) == 36 and set(res[0][1]) == set(res[1][1])
    for k, a in res[0][1].items():
        b = res[1][1][k]
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos >= 0.9999, (k, cos)          # measured at B=32 256x256: worst relative L2 difference 6.4e-3 (cos 0.99998)
