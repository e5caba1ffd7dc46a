"""GPU parity of every HIP kernel against stock-PyTorch CPU fp32 (the oracle's arithmetic), through the
C ABI (ctypes -> libwu_kernels.so).  fp32 kernels: tight tolerances (exact-fp32 MFMA, only the summation
order differs).  bf16 kernels: the reference is evaluated on bf16-ROUNDED inputs/weights, so what remains
is accumulation order + one output rounding (2^-9 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = ["fp32", "bf16"]


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _tdt(p):
    return torch.float32 if p == "fp32" else torch.bfloat16


def _round(t, p):
    return t.to(_tdt(p)).float()


def _tol(p, ref, k=1.0):
    scale = max(1.0, float(ref.abs().max()))
    return (2e-4 if p == "fp32" else 1.2e-2) * scale * k


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def _nhwc(x_cpu, p):
    from wu.layout import as_nhwc, precision_code
    return as_nhwc(x_cpu.to(_dev()), precision_code(p))


CONV_SHAPES = [
    # N, Cin, Cout, H, W, stride
    (2, 64, 64, 16, 32, 1),
    (1, 64, 128, 24, 40, 1),      # ragged: H, W not multiples of the 8x32 tile
    (2, 192, 64, 8, 8, 1),        # narrow image -> TW = 8 tiles; concat-sized Cin
    (1, 128, 128, 4, 4, 1),       # 4x4 bottleneck of a 32x32 input
    (3, 64, 64, 33, 35, 1),       # odd sizes
    (2, 192, 128, 40, 72, 1),     # several tiles per split in the LDS-DMA wgrad, ragged in both directions
    (1, 256, 128, 20, 40, 1),     # Cin >= 256: the one-wave-per-SIMD (NW = 4) conv shape, 8 chunks per tile, ragged
    (2, 384, 64, 16, 64, 1),      # NW = 4, 12 chunks, W % 32 == 0 (LDS-DMA wgrad eligible), 6 ci-blocks
    (2, 64, 128, 16, 32, 2),      # discriminator stride-2
    (1, 128, 64, 10, 12, 2),
]


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv3x3_forward(p, shape, act):
    from wu import functional as WF
    n, cin, cout, h, w, stride = shape
    if act == 2 and stride == 1 and cin not in (64, 256):      # 64: the 8-wave LDS-DMA instance, 256: the 4-wave 16x16x32 one
        pytest.skip("leaky covered on a subset")
    x = _round(_rand((n, cin, h, w), 1), p)
    wt = _round(_rand((cout, cin, 3, 3), 2, -0.1, 0.1), p)
    b = _rand((cout,), 3)
    ref = F.conv2d(x, wt, b, stride=stride, padding=1)
    ref = {0: ref, 1: F.relu(ref), 2: F.leaky_relu(ref, 0.2)}[act]
    y = WF.conv3x3(_nhwc(x, p), wt.to(_dev()), b.to(_dev()), WF.PackedConv(), stride, act)
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= _tol(p, ref), f"max-abs {err}"


@pytest.mark.parametrize("p", DTYPES)
def test_conv3x3_into_concat_slice(p):
    """Output written into a channel slice of a wider NHWC buffer (the zero-copy torch.cat, cunet.py:62)."""
    from wu import functional as WF
    from wu.layout import empty_nhwc
    n, cin, cout, h, w = 2, 64, 64, 16, 32
    x = _round(_rand((n, cin, h, w), 4), p)
    wt = _round(_rand((cout, cin, 3, 3), 5, -0.1, 0.1), p)
    b = _rand((cout,), 6)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    cat = empty_nhwc(n, 128 + cout, h, w, _tdt(p), _dev())
    cat.fill_(7.0)
    y = WF.conv3x3(_nhwc(x, p), wt.to(_dev()), b.to(_dev()), WF.PackedConv(), 1, 1, cat[:, 128:])
    assert y.data_ptr() == cat[:, 128:].data_ptr()
    assert (cat[:, 128:].float().cpu() - ref).abs().max().item() <= _tol(p, ref)
    assert (cat[:, :128].float() == 7.0).all()       # neighbours untouched
    # and read back as a strided input
    y2 = WF.conv3x3(cat[:, 128:], wt.to(_dev()), b.to(_dev()), WF.PackedConv(), 1, 0)
    ref2 = F.conv2d(_round(ref, p), wt, b, padding=1)
    assert (y2.float().cpu() - ref2).abs().max().item() <= _tol(p, ref2, 2.0)


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv3x3_backward(p, shape, act):
    """dgrad (gated-input MFMA pass / zero-upsampled stride-2 pass) and wgrad (transposed-LDS-read MFMA +
    deterministic split-K) against torch autograd on CPU."""
    from wu import functional as WF
    n, cin, cout, h, w, stride = shape
    if act == 2 and cin != 64:
        pytest.skip("leaky covered on a subset")
    x = _round(_rand((n, cin, h, w), 11), p).requires_grad_(True)
    wt = _round(_rand((cout, cin, 3, 3), 12, -0.1, 0.1), p).requires_grad_(True)
    b = _rand((cout,), 13).requires_grad_(True)
    pre = F.conv2d(x, wt, b, stride=stride, padding=1)
    ref = {0: pre, 1: F.relu(pre), 2: F.leaky_relu(pre, 0.2)}[act]
    gy = _round(_rand(tuple(ref.shape), 14), p)
    # the kernel gates on the STORED (rounded) output; make the reference gate identically
    ref.backward(gy)
    xg = _nhwc(x.detach(), p).requires_grad_(True)
    wg = wt.detach().to(_dev()).requires_grad_(True)
    bg = b.detach().to(_dev()).requires_grad_(True)
    y = WF.conv3x3(xg, wg, bg, WF.PackedConv(), stride, act)
    y.backward(_nhwc(gy, p))
    for name, got, want, k in (("dx", xg.grad, x.grad, 1.0), ("dw", wg.grad, wt.grad, 4.0), ("db", bg.grad, b.grad, 4.0)):
        err = (got.float().cpu() - want).abs().max().item()
        tol = _tol(p, want, k)
        if name != "dx" and p == "bf16":
            tol = 2e-3 * float(want.abs().max()) + 1e-3     # fp32-accumulated: tighter than the bf16-stored dx
        assert err <= tol, f"{name}: max-abs {err} (tol {tol}, |ref|max {want.abs().max().item()})"


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("act", [1, 2])
def test_conv3x3_gated_abi_paths(p, act):
    """C ABI: the in-kernel gated variants (mask / y arguments) equal gate-then-GEMM through wu_act_gate."""
    from wu import _lib, functional as WF
    from wu.layout import empty_nhwc, nhwc_ld, precision_code, stream_ptr
    code = precision_code(p)
    n, cin, cout, h, w = 2, 64, 128, 12, 40
    x = _nhwc(_round(_rand((n, cin, h, w), 71), p), p)
    y = _nhwc(_round(_rand((n, cout, h, w), 72), p), p)          # stands for the stored activation
    gy = _nhwc(_round(_rand((n, cout, h, w), 73), p), p)
    wt = _round(_rand((cout, cin, 3, 3), 74, -0.1, 0.1), p).to(_dev())
    w_fwd, w_dgrad = WF.PackedConv().get(wt, code)
    s = stream_ptr()
    gg = empty_nhwc(n, cout, h, w, _tdt(p), _dev())
    _lib.call("wu_act_gate", gy.data_ptr(), nhwc_ld(gy), y.data_ptr(), nhwc_ld(y), gg.data_ptr(), nhwc_ld(gg), n, h, w, cout, act, code, s)
    ref_gate = gy.float() * torch.where(y.float() > 0, torch.ones(()), torch.full((), 0.0 if act == 1 else 0.2)).to(_dev())
    assert (gg.float() - ref_gate).abs().max().item() <= (0 if p == "fp32" else 4e-3)
    dx = [empty_nhwc(n, cin, h, w, _tdt(p), _dev()) for _ in range(2)]
    _lib.call("wu_conv3x3_fwd", gy.data_ptr(), nhwc_ld(gy), w_dgrad.data_ptr(), None, dx[0].data_ptr(), cin, n, h, w, cout, cin, 1, 0,
              y.data_ptr(), nhwc_ld(y), act, None, 0, 0, code, s)
    _lib.call("wu_conv3x3_fwd", gg.data_ptr(), nhwc_ld(gg), w_dgrad.data_ptr(), None, dx[1].data_ptr(), cin, n, h, w, cout, cin, 1, 0,
              None, 0, 0, None, 0, 0, code, s)
    assert torch.equal(dx[0], dx[1])
    # epilogue gate: output * act'(egate) == act_gate(output, egate)
    dxe = empty_nhwc(n, cin, h, w, _tdt(p), _dev())
    _lib.call("wu_conv3x3_fwd", gg.data_ptr(), nhwc_ld(gg), w_dgrad.data_ptr(), None, dxe.data_ptr(), cin, n, h, w, cout, cin, 1, 0,
              None, 0, 0, x.data_ptr(), nhwc_ld(x), act, code, s)
    ref_e = empty_nhwc(n, cin, h, w, _tdt(p), _dev())
    _lib.call("wu_act_gate", dx[1].data_ptr(), cin, x.data_ptr(), nhwc_ld(x), ref_e.data_ptr(), cin, n, h, w, cin, act, code, s)
    assert torch.equal(dxe, ref_e)
    nbytes = _lib.load().wu_conv3x3_wgrad_workspace(n, h, w, cin, cout, 1, code)
    ws = WF.workspace(nbytes, _dev())
    dw = [torch.empty((cout, cin, 3, 3), device=_dev()) for _ in range(2)]
    db = [torch.empty((cout,), device=_dev()) for _ in range(2)]
    _lib.call("wu_conv3x3_wgrad", x.data_ptr(), cin, gy.data_ptr(), cout, y.data_ptr(), cout, act, dw[0].data_ptr(), db[0].data_ptr(),
              ws.data_ptr(), ws.numel(), n, h, w, cin, cout, 1, 0, code, s)       # gated in-kernel (generic kernel)
    _lib.call("wu_conv3x3_wgrad", x.data_ptr(), cin, gg.data_ptr(), cout, None, 0, 0, dw[1].data_ptr(), db[1].data_ptr(),
              ws.data_ptr(), ws.numel(), n, h, w, cin, cout, 1, 0, code, s)       # pre-gated (LDS-DMA kernel for bf16)
    tol = 1e-4 if p == "fp32" else 2e-3
    assert (dw[0] - dw[1]).abs().max().item() <= tol * max(1.0, dw[1].abs().max().item())
    assert (db[0] - db[1]).abs().max().item() <= tol * max(1.0, db[1].abs().max().item())
    # accumulate != 0 adds into the existing gradient (autograd's .grad accumulation)
    _lib.call("wu_conv3x3_wgrad", x.data_ptr(), cin, gg.data_ptr(), cout, None, 0, 0, dw[1].data_ptr(), db[1].data_ptr(),
              ws.data_ptr(), ws.numel(), n, h, w, cin, cout, 1, 1, code, s)
    assert (dw[1] - 2 * dw[0]).abs().max().item() <= 2 * tol * max(1.0, dw[0].abs().max().item())


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 64, 16, 24, 1, 1, False), (2, 64, 17, 19, 2, 2, False), (2, 3, 12, 20, 1, 0, True),
                                 (2, 64, 40, 72, 2, 2, False), (1, 64, 34, 130, 2, 0, False),
                                 # the LDS-tiled image-layout 3 -> 3 conv: ragged tiles (W % 4 != 0), several tiles, an activation
                                 (2, 3, 37, 131, 1, 0, True), (1, 3, 64, 128, 1, 2, True), (3, 3, 16, 64, 1, 1, True)])
def test_conv3x3_c3(p, cfg):
    """3-input-channel conv read from the NCHW fp32 image (cunet.py:45, disc.py:28): fwd, wgrad, dgrad."""
    from wu import functional as WF
    from wu.layout import precision_code
    n, cout, h, w, stride, act, out_nchw = cfg
    x = _rand((n, 3, h, w), 21).requires_grad_(True)
    wt = _rand((cout, 3, 3, 3), 22, -0.3, 0.3).requires_grad_(True)
    b = _rand((cout,), 23).requires_grad_(True)
    pre = F.conv2d(x, wt, b, stride=stride, padding=1)
    ref = {0: pre, 1: F.relu(pre), 2: F.leaky_relu(pre, 0.2)}[act]
    gy = _round(_rand(tuple(ref.shape), 24), "fp32" if out_nchw else p)
    ref.backward(gy)
    xg = x.detach().to(_dev()).requires_grad_(True)
    wg = wt.detach().to(_dev()).requires_grad_(True)
    bg = b.detach().to(_dev()).requires_grad_(True)
    y = WF.conv3x3_c3(xg, wg, bg, stride, act, out_nchw, precision_code(p))
    pp = "fp32" if out_nchw else p
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= _tol(pp, ref.detach())
    y.backward(gy.to(_dev()) if out_nchw else _nhwc(gy, p))
    # fp32: exact gating, tight max-abs.  bf16: the forward runs on the matrix cores with bf16-rounded image and weights, so
    # pre-activations within ~1e-2 of zero can land on the other side of the ReLU / LeakyReLU kink than in the fp32
    # reference -- the same gate-flip noise every bf16 layer has: relative-L2 bound instead of max-abs
    for name, got, want in (("dx", xg.grad, x.grad), ("dw", wg.grad, wt.grad), ("db", bg.grad, b.grad)):
        d = got.float().cpu() - want
        if pp == "bf16":
            rel = (d.norm() / want.norm()).item()
            assert rel <= 4e-2, f"{name}: rel-L2 {rel}"
        else:
            err = d.abs().max().item()
            assert err <= 2e-4 * max(1.0, float(want.abs().max())), f"{name}: {err}"


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 16, 32), (1, 128, 6, 10), (3, 256, 4, 4)])
def test_maxpool(p, shape):
    from wu import functional as WF
    n, c, h, w = shape
    x = _round(_rand((n, c, h, w), 31), p)
    # force ties (bf16-like plateaus) so the first-max tie rule is exercised
    x = (x * 4).round() / 4
    x.requires_grad_(True)
    ref = F.max_pool2d(x, 2)
    gy = _round(_rand(tuple(ref.shape), 32), p)
    ref.backward(gy)
    xg = _nhwc(x.detach(), p).requires_grad_(True)
    y = WF.maxpool2(xg)
    assert torch.equal(y.float().cpu(), ref.detach())
    y.backward(_nhwc(gy, p))
    assert torch.equal(xg.grad.float().cpu(), x.grad)


@pytest.mark.parametrize("p", DTYPES)
def test_maxpool_bwd_fused_skip(p):
    """C ABI: dx = route(dy) + dskip in one pass (the encoder's two-consumer gradient sum)."""
    from wu import _lib
    from wu.layout import empty_nhwc, nhwc_ld, precision_code, stream_ptr
    n, c, h, w = 2, 64, 8, 16
    x = _round(_rand((n, c, h, w), 33), p).requires_grad_(True)
    ref = F.max_pool2d(x, 2)
    gy = _round(_rand(tuple(ref.shape), 34), p)
    gs = _round(_rand((n, c, h, w), 35), p)
    ref.backward(gy)
    want = x.grad + gs
    xd, gyd, gsd = _nhwc(x.detach(), p), _nhwc(gy, p), _nhwc(gs, p)
    dx = empty_nhwc(n, c, h, w, _tdt(p), _dev())
    _lib.call("wu_maxpool2_bwd", xd.data_ptr(), nhwc_ld(xd), gyd.data_ptr(), nhwc_ld(gyd), gsd.data_ptr(), nhwc_ld(gsd),
              dx.data_ptr(), nhwc_ld(dx), n, h, w, c, 0, precision_code(p), stream_ptr())
    assert (dx.float().cpu() - want).abs().max().item() <= (1e-6 if p == "fp32" else 1.6e-2)
    # ... and additionally gated by ReLU'(x) (the fused-graph form)
    _lib.call("wu_maxpool2_bwd", xd.data_ptr(), nhwc_ld(xd), gyd.data_ptr(), nhwc_ld(gyd), gsd.data_ptr(), nhwc_ld(gsd),
              dx.data_ptr(), nhwc_ld(dx), n, h, w, c, 1, precision_code(p), stream_ptr())
    want_g = want * (x.detach() > 0)
    assert (dx.float().cpu() - want_g).abs().max().item() <= (1e-6 if p == "fp32" else 1.6e-2)


def _adain_upcat_ref(x, c_std, c_mean, skip, eps, mask):
    n, c = x.shape[:2]
    xf = x.reshape(n, c, -1)
    x_std = (xf.var(dim=-1) + eps).sqrt().view(n, c, 1, 1)
    x_mean = xf.mean(dim=-1).view(n, c, 1, 1)
    y = (x - x_mean) / x_std * c_std.view(n, c, 1, 1) + c_mean.view(n, c, 1, 1)
    y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
    if mask is not None:
        y = y * mask / 0.7
    return torch.cat([y, skip], dim=1)


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("shape", [(2, 128, 8, 8, 64), (1, 256, 4, 6, 128), (2, 64, 16, 16, 64), (1, 64, 5, 7, 64), (2, 128, 40, 72, 64)])
@pytest.mark.parametrize("train", [False, True])
def test_adain_upcat(p, shape, train):
    """AdaIN (utils.py:41-51) -> bilinear x2 align_corners (cunet.py:26) -> Dropout(0.3) -> cat, fused; forward
    and backward (dx, d y_std, d y_mean, d skip).  Train mode: the oracle gets the kernel's own keep-mask."""
    from wu import functional as WF
    from wu.layout import empty_nhwc
    n, c, h, w, cs = shape
    eps = 1e-5
    x = _round(_rand((n, c, h, w), 41, -1, 2), p).requires_grad_(True)
    ystd = _rand((n, c), 42, 0.5, 1.5).requires_grad_(True)
    ymean = _rand((n, c), 43).requires_grad_(True)
    skip = _round(_rand((n, cs, 2 * h, 2 * w), 44), p).requires_grad_(True)
    seed = 1234567
    mask = WF.dropout_mask(n, c, 2 * h, 2 * w, 0.3, seed, _dev()).float().cpu() if train else None
    if train:
        assert abs(mask.mean().item() - 0.7) < 0.02
    ref = _adain_upcat_ref(x, ystd, ymean, skip, eps, mask)
    g = _round(_rand(tuple(ref.shape), 45), p)
    ref.backward(g)

    xg = _nhwc(x.detach(), p).requires_grad_(True)
    sg, mg = ystd.detach().to(_dev()).requires_grad_(True), ymean.detach().to(_dev()).requires_grad_(True)
    cat = empty_nhwc(n, c + cs, 2 * h, 2 * w, _tdt(p), _dev())
    cat[:, c:].copy_(skip.detach().to(_dev()))
    skg = cat[:, c:].detach().requires_grad_(True)
    out = WF.adain_upcat(xg, sg, mg, skg, cat, eps, 0.3 if train else 0.0, seed)
    err = (out.float().cpu() - ref.detach()).abs().max().item()
    assert err <= _tol(p, ref.detach(), 2.0), f"fwd {err}"
    out.backward(_nhwc(g, p))
    k = {"fp32": 2e-4, "bf16": 3e-2}[p]
    for name, got, want in (("dx", xg.grad, x.grad), ("dstd", sg.grad, ystd.grad), ("dmean", mg.grad, ymean.grad), ("dskip", skg.grad, skip.grad)):
        err = (got.float().cpu() - want).abs().max().item()
        assert err <= k * max(1.0, float(want.abs().max())), f"{name}: {err} vs |ref| {want.abs().max().item()}"


@pytest.mark.parametrize("p", DTYPES)
def test_adain_upcat_bwd_mask_bits_vs_rehash(p):
    """C ABI: backward reading the stored keep-bits == backward regenerating the mask from the seed."""
    from wu import _lib, functional as WF
    from wu.layout import empty_nhwc, nhwc_ld, precision_code, stream_ptr
    code = precision_code(p)
    n, c, h, w, cs, seed = 2, 128, 6, 10, 64, 424242
    x = _nhwc(_round(_rand((n, c, h, w), 81, -1, 2), p), p)
    g = _nhwc(_round(_rand((n, c + cs, 2 * h, 2 * w), 82), p), p)
    ystd = _rand((n, c), 83, 0.5, 1.5).to(_dev())
    ymean = _rand((n, c), 84).to(_dev())
    stats = WF.adain_stats(x, 1e-5)
    cat = empty_nhwc(n, c + cs, 2 * h, 2 * w, _tdt(p), _dev())
    esz = 4 if p == "fp32" else 2
    bits = torch.zeros(n * 4 * h * w * (c * esz // 16), dtype=torch.uint8, device=_dev())
    s = stream_ptr()
    _lib.call("wu_adain_upcat_fwd", x.data_ptr(), nhwc_ld(x), stats.data_ptr(), ystd.data_ptr(), ymean.data_ptr(),
              cat.data_ptr(), nhwc_ld(cat), n, h, w, c, 0.3, seed, None, bits.data_ptr(), 0, code, s)
    keep_frac = sum(bin(int(b)).count("1") for b in bits[:4096].cpu().tolist()) / (4096 * (16 // esz))
    assert abs(keep_frac - 0.7) < 0.03
    res = []
    # both runs on the MARCHING kernel (option 8 = 5): the LDS-ring kernel of round 4 needs stored bits, and its per-(n, c) partial sums
    # are grouped by other tiles (tests/test_gpu_round4.py compares it with this one)
    _lib.call("wu_set_option", 8, 5)
    for mb in (bits.data_ptr(), None):
        dx = empty_nhwc(n, c, h, w, _tdt(p), _dev())
        dstd, dmean = torch.empty((n, c), device=_dev()), torch.empty((n, c), device=_dev())
        gtmp = torch.empty((n, h, w, c), dtype=_tdt(p), device=_dev())
        sums = torch.empty((n, c, 2 * (1 + WF.MAX_SPLITS)), device=_dev())
        _lib.call("wu_adain_upcat_bwd", g.data_ptr(), nhwc_ld(g), x.data_ptr(), nhwc_ld(x), stats.data_ptr(), ystd.data_ptr(),
                  dx.data_ptr(), nhwc_ld(dx), dstd.data_ptr(), dmean.data_ptr(), gtmp.data_ptr(), sums.data_ptr(),
                  n, h, w, c, 0.3, seed, mb, 0, code, s)
        res.append((dx.clone(), dstd.clone(), dmean.clone()))
    _lib.call("wu_set_option", 8, 1)
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("p", DTYPES)
def test_conv1x1_tanh(p):
    from wu import functional as WF
    n, cin, h, w = 2, 64, 12, 20
    x = _round(_rand((n, cin, h, w), 51), p).requires_grad_(True)
    wt = _rand((3, cin, 1, 1), 52, -0.3, 0.3).requires_grad_(True)
    b = _rand((3,), 53).requires_grad_(True)
    ref = torch.tanh(F.conv2d(x, wt, b))
    g = _rand(tuple(ref.shape), 54)
    ref.backward(g)
    xg = _nhwc(x.detach(), p).requires_grad_(True)
    wg, bg = wt.detach().to(_dev()).requires_grad_(True), b.detach().to(_dev()).requires_grad_(True)
    out = WF.conv1x1_tanh(xg, wg, bg)
    assert out.dtype == torch.float32 and out.is_contiguous()
    assert (out.cpu() - ref.detach()).abs().max().item() <= 2e-5
    out.backward(g.to(_dev()))
    assert (xg.grad.float().cpu() - x.grad).abs().max().item() <= (1e-5 if p == "fp32" else 8e-3)
    assert (wg.grad.cpu() - wt.grad).abs().max().item() <= 1e-3 * max(1.0, float(wt.grad.abs().max()))
    assert (bg.grad.cpu() - b.grad).abs().max().item() <= 1e-3 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize("p", DTYPES)
def test_sumpool_and_layout(p):
    from wu import functional as WF
    from wu.layout import to_nchw_f32
    n, c, h, w = 2, 128, 5, 7
    x = _round(_rand((n, c, h, w), 61), p).requires_grad_(True)
    ref = x.sum(dim=[2, 3])
    g = _rand((n, c), 62)
    ref.backward(g)
    xg = _nhwc(x.detach(), p).requires_grad_(True)
    assert torch.equal(to_nchw_f32(xg.detach()).cpu(), x.detach())       # NHWC <-> NCHW round trip is exact
    f = WF.sumpool(xg)
    assert (f.cpu() - ref.detach()).abs().max().item() <= 1e-4
    f.backward(g.to(_dev()))
    assert (xg.grad.float().cpu() - x.grad).abs().max().item() <= (0 if p == "fp32" else 4e-3)


def test_rejects_bad_arguments():
    """Error behaviour of the boundary: shape/alignment violations come back as RuntimeError, CPU tensors are
    refused (no fallback)."""
    from wu import functional as WF
    x = torch.zeros(1, 40, 8, 8, device=_dev()).contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="Cin"):
        WF.conv3x3(x, torch.zeros(64, 40, 3, 3, device=_dev()), None, WF.PackedConv(), 1, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        WF.maxpool2(torch.zeros(1, 64, 8, 8))


def test_adain_style_fused():
    """utils.py:41-48: l1(y).view(N, C, 4) -> (sqrt(unbiased var + eps), mean) and its gradients into l1, one kernel each,
    against stock autograd (fp32)."""
    from wu import functional as WF
    n, c, nc, eps = 6, 40, 5, 1e-5
    y = _rand((n, nc), 71)
    w = _rand((4 * c, nc), 72).requires_grad_(True)
    b = _rand((4 * c,), 73).requires_grad_(True)
    y_ = F.linear(y, w, b).view(n, c, 4)
    ref_std, ref_mean = (y_.var(dim=-1) + eps).sqrt(), y_.mean(dim=-1)
    gs, gm = _rand((n, c), 74), _rand((n, c), 75)
    (ref_std * gs + ref_mean * gm).sum().backward()
    wg = w.detach().to(_dev()).requires_grad_(True)
    bg = b.detach().to(_dev()).requires_grad_(True)
    std, mean = WF.adain_style(y.to(_dev()), wg, bg, eps)
    assert (std.cpu() - ref_std.detach()).abs().max().item() <= 1e-5 and (mean.cpu() - ref_mean.detach()).abs().max().item() <= 1e-5
    (std * gs.to(_dev()) + mean * gm.to(_dev())).sum().backward()
    assert (wg.grad.cpu() - w.grad).abs().max().item() <= 2e-5 * max(1.0, float(w.grad.abs().max()))
    assert (bg.grad.cpu() - b.grad).abs().max().item() <= 2e-5 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize("p", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 64, 16, 32), (1, 64, 128, 24, 40), (3, 64, 64, 34, 38), (1, 256, 128, 20, 40), (2, 192, 64, 8, 8)])
def test_conv3x3_relu_pool_fused(p, shape):
    """cunet.py:45-46 etc.: the encoder conv's epilogue also writes max_pool2d(y, 2) (LDS-DMA bf16 path; conv + pool kernel
    otherwise): y must equal the plain conv bit for bit and the pooled tensor must be the exact 2x2 maximum of that y."""
    from wu import functional as WF, kernels as K
    from wu.layout import empty_nhwc, precision_code, torch_dtype
    n, cin, cout, h, w = shape
    code = precision_code(p)
    x = _round(_rand((n, cin, h, w), 81), p)
    wt = _round(_rand((cout, cin, 3, 3), 82, -0.1, 0.1), p)
    b = _rand((cout,), 83)
    xg, bg = _nhwc(x, p), b.to(_dev())
    wf, _ = K.pack_conv3x3(wt.to(_dev()), code)
    dt = torch_dtype(code)
    y_plain = K.conv3x3(xg, wf, bg, empty_nhwc(n, cout, h, w, dt, xg.device), 1, 1)
    y = empty_nhwc(n, cout, h, w, dt, xg.device)
    pool = empty_nhwc(n, cout, h // 2, w // 2, dt, xg.device)
    K.conv3x3_relu_pool(xg, wf, bg, y, pool)
    assert torch.equal(y, y_plain)
    ref_pool = F.max_pool2d(y_plain.float(), 2)
    assert torch.equal(pool.float(), ref_pool)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    assert (y.float().cpu() - ref).abs().max().item() <= _tol(p, ref)


@pytest.mark.parametrize("p", DTYPES)
def test_adain_upcat_bwd_marching_vs_gather(p):
    """The two formulations of the fused AdaIN / bilinear / dropout backward (16-tap gather per low-res pixel; separable
    "marching" over high-res rows, two low-res columns per thread) compute the same sums in a different order."""
    from wu import _lib, functional as WF, kernels as K
    from wu.layout import empty_nhwc
    n, c, h, w, cs, seed = 2, 128, 24, 40, 64, 99
    x = _nhwc(_round(_rand((n, c, h, w), 81, -1, 2), p), p)
    g = _nhwc(_round(_rand((n, c + cs, 2 * h, 2 * w), 82), p), p)
    ystd = _rand((n, c), 83, 0.5, 1.5).to(_dev())
    ymean = _rand((n, c), 84).to(_dev())
    stats = K.adain_stats(x, 1e-5)
    cat = empty_nhwc(n, c + cs, 2 * h, 2 * w, _tdt(p), _dev())
    mb = K.adain_upcat(x, stats, ystd, ymean, cat, 0.3, seed, True)
    res = {}
    try:
        for mode in (0, 1):
            _lib.call("wu_set_option", 8, 5 if mode == 1 else 0)      # 5 = the marching kernel also where the LDS-ring kernel would run
            for bits in (mb, None):
                dx = empty_nhwc(n, c, h, w, _tdt(p), _dev())
                ds, dm = K.adain_upcat_bwd(g, x, stats, ystd, dx, 0.3, seed, bits, 1)
                res[(mode, bits is None)] = (dx.float().clone(), ds.clone(), dm.clone())
    finally:
        _lib.call("wu_set_option", 8, 1)
    for k in ((1, False), (1, True), (0, True)):
        for a_, b_ in zip(res[k], res[(0, False)]):
            scale = max(1.0, b_.abs().max().item())
            assert (a_ - b_).abs().max().item() <= (1e-4 if p == "fp32" else 2e-2) * scale, k
    assert torch.equal(res[(1, False)][0], res[(1, True)][0])        # stored keep-bits == regenerated mask


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (3, 5, 7), (1, 3, 64, 64)])
def test_l1_loss_fused(shape):
    """ops.l1_loss on fp32 device tensors (reference ops.py:22-24): value and both gradients against the torch op, including an
    element count that is not a multiple of 4, exact zeros in a - b (sign 0) and a non-unit upstream gradient."""
    import ops
    a = _rand(shape, 71)
    b = _rand(shape, 72)
    b.view(-1)[::7] = a.view(-1)[::7]                    # exact ties: sign(0) = 0
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.l1_loss(ar, br) * 3.0).backward()
    ag, bg = a.to(_dev()).requires_grad_(True), b.to(_dev()).requires_grad_(True)
    loss = ops.l1_loss(ag, bg)
    (loss * 3.0).backward()
    assert abs(loss.item() - F.l1_loss(a, b).item()) <= 1e-6 * max(1.0, F.l1_loss(a, b).item())
    # sign(a - b) * (3 / n): the scale may differ from torch's by one rounding (1/n * 3 vs 3/n)
    assert torch.allclose(ag.grad.cpu(), ar.grad, rtol=1e-6, atol=0) and torch.allclose(bg.grad.cpu(), br.grad, rtol=1e-6, atol=0)
    assert torch.equal(ag.grad.cpu() == 0, ar.grad == 0)
    # run-to-run identical (fixed summation order)
    assert ops.l1_loss(ag.detach(), bg.detach()).item() == ops.l1_loss(ag.detach(), bg.detach()).item()


def _decode_gate_bits(bits, n, c, h, w):
    """uint32 [N*H*W][C/64][2] -> bool (N,C,H,W): bit 8k+i of word (p, ct, hf) is channel 64ct + 16k + 8hf + i (wu_kernels.h)."""
    b = bits.cpu().numpy().view("uint32").reshape(n, h, w, c // 64, 2)
    out = torch.zeros((n, c, h, w), dtype=torch.bool)
    for ct in range(c // 64):
        for hf in range(2):
            word = b[:, :, :, ct, hf]
            for k in range(4):
                for i in range(8):
                    out[:, 64 * ct + 16 * k + 8 * hf + i] = torch.from_numpy(((word >> (8 * k + i)) & 1).astype("bool"))
    return out


@pytest.mark.parametrize("shape", [(2, 64, 64, 16, 32), (1, 64, 128, 24, 40), (2, 256, 128, 20, 40), (1, 128, 64, 33, 35)])
def test_conv3x3_gate_bits(shape):
    """ReLU gates carried as bits: the forward writes exactly (y > 0) next to an unchanged y; the data-gradient pass gated by
    the bits equals the one gated by the tensor bit for bit (both conv3x3_mfma_v2 instances, both wave counts, ragged tiles)."""
    from wu import kernels as K
    from wu.layout import empty_nhwc, precision_code
    n, cin, cout, h, w = shape
    p = "bf16"
    x = _nhwc(_round(_rand((n, cin, h, w), 81), p), p)
    wt = _round(_rand((cout, cin, 3, 3), 82, -0.1, 0.1), p).to(_dev())
    b = _rand((cout,), 83).to(_dev())
    wf, wd = K.pack_conv3x3(wt, precision_code(p))
    y0 = empty_nhwc(n, cout, h, w, x.dtype, x.device)
    y1 = empty_nhwc(n, cout, h, w, x.dtype, x.device)
    assert K.gate_bits_supported(x, y0)
    K.conv3x3(x, wf, b, y0, 1, 1)
    bits = K.gate_bits_alloc(y1)
    bits.fill_(-1)
    K.conv3x3_bits(x, wf, b, y1, 1, gate_bits_out=bits)
    assert torch.equal(y0, y1)
    assert torch.equal(_decode_gate_bits(bits, n, cout, h, w), (y0.float() > 0).cpu())
    # data gradient of this conv's OUTPUT-side neighbour: dx = conv(gy, w_dgrad) gated by ReLU'(x_prev) with x_prev := some tensor
    gy = _nhwc(_round(_rand((n, cout, h, w), 84), p), p)
    prev = _nhwc(torch.relu(_round(_rand((n, cin, h, w), 85), p)), p)          # the "mid" activation whose gate is applied (Cin channels)
    pbits = K.gate_bits_alloc(prev)
    # bits of `prev` through the same writer: a forward conv that reproduces prev is not available, so build them on the host
    pb = torch.zeros((n, h, w, cin // 64, 2), dtype=torch.int64)
    pos = (prev.float() > 0).cpu()
    for ct in range(cin // 64):
        for hf in range(2):
            for k in range(4):
                for i in range(8):
                    pb[:, :, :, ct, hf] |= pos[:, 64 * ct + 16 * k + 8 * hf + i].long() << (8 * k + i)
    pb = torch.where(pb >= 2 ** 31, pb - 2 ** 32, pb).to(torch.int32)
    pbits.copy_(pb.reshape(-1).to(_dev()))
    d0 = empty_nhwc(n, cin, h, w, x.dtype, x.device)
    d1 = empty_nhwc(n, cin, h, w, x.dtype, x.device)
    K.conv3x3(gy, wd, None, d0, 1, 0, egate=prev, egate_act=1)
    K.conv3x3_bits(gy, wd, None, d1, 0, egate_bits=pbits)
    assert torch.equal(d0, d1)


def test_conv3x3_c3_gate_bits():
    from wu import kernels as K
    from wu.layout import empty_nhwc, precision_code
    n, h, w = 2, 24, 40
    x = _rand((n, 3, h, w), 91).to(_dev())
    wt = _rand((64, 3, 3, 3), 92, -0.3, 0.3).to(_dev())
    b = _rand((64,), 93).to(_dev())
    code = precision_code("bf16")
    y0 = empty_nhwc(n, 64, h, w, torch.bfloat16, x.device)
    y1 = empty_nhwc(n, 64, h, w, torch.bfloat16, x.device)
    assert K.conv3x3_c3_bits_supported(x, wt, b, 1, code)
    K.conv3x3_c3(x, wt, b, y0, 1, 1, False, code)
    bits = K.gate_bits_alloc(y1)
    bits.fill_(-1)
    K.conv3x3_c3_bits(x, wt, b, y1, bits, 1, code)
    assert torch.equal(y0, y1)
    assert torch.equal(_decode_gate_bits(bits, n, 64, h, w), (y0.float() > 0).cpu())


@pytest.mark.parametrize("p", DTYPES)
def test_pack_conv3x3_multi_equals_single(p):
    """The batched repack (one launch for all stale weights after an optimizer step) writes the same two operand images as the
    per-weight kernel, ragged channel counts included."""
    from wu import kernels as K
    from wu.layout import precision_code
    code = precision_code(p)
    ws = [_rand(shape, 300 + i, -0.2, 0.2).to(_dev()) for i, shape in enumerate(
        [(64, 64, 3, 3), (128, 64, 3, 3), (64, 192, 3, 3), (256, 768, 3, 3), (48, 80, 3, 3), (16, 16, 3, 3)] + [(64, 64, 3, 3)] * 12)]
    outs = K.pack_conv3x3_multi(ws, code)          # 18 weights: two launches
    for w, (wf, wd) in zip(ws, outs):
        rf, rd = K.pack_conv3x3(w, code)
        assert torch.equal(wf, rf) and torch.equal(wd, rd)
