"""Typed launch helpers over the C ABI (include/wu_kernels.h): tensors in, one ``_lib.call`` out.

Every helper takes NHWC-strided tensors (logical NCHW shape, see ``layout``), passes (pointer, pixel stride)
pairs through ctypes and enqueues on torch's current HIP stream.  Nothing here allocates except the
explicitly named workspaces; nothing here falls back to torch arithmetic.
"""
import torch

from . import _lib
from .layout import dtype_code, empty_nhwc, nhwc_ld, stream_ptr

ACT_NONE, ACT_RELU, ACT_LEAKY = _lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_LEAKY
MAX_SPLITS = 16     # WU_MAX_SPLITS

_WS = {}


def workspace(nbytes, device):
    """One growable caller-owned byte buffer per device AND stream (the C ABI never allocates; kernels enqueued on different
    streams must not share or regrow each other's scratch)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _pl(t):
    """(pointer, ld) of an optional NHWC tensor."""
    return (t.data_ptr(), nhwc_ld(t)) if t is not None else (None, 0)


def conv3x3(x, w_packed, bias, y, stride=1, act=ACT_NONE, mask=None, mask_act=ACT_NONE, egate=None, egate_act=ACT_NONE):
    """y = act(conv3x3(x [gated by mask]) + bias) [* act'(egate)]  -- wu_conv3x3_fwd."""
    n, cin, h, w = x.shape
    cout = y.shape[1]
    mp, mld = _pl(mask)
    ep, eld = _pl(egate)
    _lib.call("wu_conv3x3_fwd", x.data_ptr(), nhwc_ld(x), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), nhwc_ld(y), n, h, w, cin, cout, stride, act, mp, mld, mask_act, ep, eld, egate_act,
              dtype_code(x), stream_ptr())
    return y


def gate_bits_supported(x, y):
    """The LDS-DMA conv (the only one with the gate-bits epilogue) takes x -> y (both NHWC views)."""
    n, cin, h, w = x.shape
    return bool(_lib.load().wu_conv3x3_gate_bits_supported(h, w, nhwc_ld(x), nhwc_ld(y), cin, y.shape[1], dtype_code(x)))


def gate_bits_alloc(y):
    """A bits buffer for activation tensor y (wu_kernels.h, "gate bits": uint32 [N*H*W][C/64][2])."""
    n, c, h, w = y.shape
    return torch.empty(_lib.load().wu_gate_bits_bytes(n, h, w, c) // 4, dtype=torch.int32, device=y.device)


def conv3x3_bits(x, w_packed, bias, y, act=ACT_NONE, gate_bits_out=None, egate_bits=None):
    """wu_conv3x3_fwd_bits: the stride-1 bf16 conv that writes (forward, ReLU) or reads (data gradient) the ReLU gate as bits."""
    n, cin, h, w = x.shape
    _lib.call("wu_conv3x3_fwd_bits", x.data_ptr(), nhwc_ld(x), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), nhwc_ld(y), gate_bits_out.data_ptr() if gate_bits_out is not None else None,
              egate_bits.data_ptr() if egate_bits is not None else None, n, h, w, cin, y.shape[1], act, dtype_code(x), stream_ptr())
    return y


def conv3x3_relu_pool(x, w_packed, bias, y, pool):
    """y = ReLU(conv3x3(x) + bias), pool = max_pool2d(y, 2) -- wu_conv3x3_relu_pool_fwd (fused epilogue on the bf16 path)."""
    n, cin, h, w = x.shape
    cout = y.shape[1]
    _lib.call("wu_conv3x3_relu_pool_fwd", x.data_ptr(), nhwc_ld(x), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), nhwc_ld(y), pool.data_ptr(), nhwc_ld(pool), n, h, w, cin, cout, dtype_code(x), stream_ptr())
    return y, pool


def conv3x3_s2_dgrad(gy, w_dgrad, dx, y=None, act=ACT_NONE, egate=None, egate_act=ACT_NONE):
    n, cin, h, w = dx.shape
    cout = gy.shape[1]
    code = dtype_code(gy)
    nbytes = _lib.load().wu_conv3x3_s2_dgrad_workspace(n, h, w, cout, code)
    ws = workspace(nbytes, gy.device)
    yp, yld = _pl(y)
    ep, eld = _pl(egate)
    _lib.call("wu_conv3x3_s2_dgrad", gy.data_ptr(), nhwc_ld(gy), yp, yld, act, w_dgrad.data_ptr(), dx.data_ptr(), nhwc_ld(dx),
              ws.data_ptr(), ws.numel(), ep, eld, egate_act, n, h, w, cin, cout, code, stream_ptr())
    return dx


def conv3x3_wgrad(x, gy, dw, db, stride=1, y=None, act=ACT_NONE, accumulate=False):
    """dw (OIHW fp32) / db from x and the (pre-gated unless y is given) output gradient gy."""
    n, cin, h, w = x.shape
    cout = gy.shape[1]
    code = dtype_code(x)
    nbytes = _lib.load().wu_conv3x3_wgrad_workspace(n, h, w, cin, cout, stride, code)
    ws = workspace(nbytes, x.device)
    yp, yld = _pl(y)
    _lib.call("wu_conv3x3_wgrad", x.data_ptr(), nhwc_ld(x), gy.data_ptr(), nhwc_ld(gy), yp, yld, act,
              dw.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), ws.numel(),
              n, h, w, cin, cout, stride, 1 if accumulate else 0, code, stream_ptr())


def act_gate(g, y, act, out=None):
    n, c, h, w = g.shape
    if out is None:
        out = empty_nhwc(n, c, h, w, g.dtype, g.device)
    _lib.call("wu_act_gate", g.data_ptr(), nhwc_ld(g), y.data_ptr(), nhwc_ld(y), out.data_ptr(), nhwc_ld(out),
              n, h, w, c, act, dtype_code(g), stream_ptr())
    return out


def conv3x3_c3(x_nchw, weight, bias, y, stride, act, out_nchw, code):
    n, _, h, w = x_nchw.shape
    cout = weight.shape[0]
    _lib.call("wu_conv3x3_c3_fwd", x_nchw.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, None,
              y.data_ptr(), 0 if out_nchw else nhwc_ld(y), 1 if out_nchw else 0, n, h, w, cout, stride, act, code, stream_ptr())
    return y


def conv3x3_c3_bits_supported(x_nchw, weight, bias, stride, code):
    n, _, h, w = x_nchw.shape
    return bool(_lib.load().wu_conv3x3_c3_gate_bits_supported(n, h, w, weight.shape[0], stride, bias.data_ptr() if bias is not None else None, code))


def conv3x3_c3_bits(x_nchw, weight, bias, y, gate_bits_out, stride, code):
    """The 3 -> 64 ReLU conv that also writes the gate bits of its output (wu_conv3x3_c3_fwd_bits)."""
    n, _, h, w = x_nchw.shape
    _lib.call("wu_conv3x3_c3_fwd_bits", x_nchw.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, None,
              y.data_ptr(), nhwc_ld(y), gate_bits_out.data_ptr(), n, h, w, weight.shape[0], stride, code, stream_ptr())
    return y


def conv3x3_c3_wgrad(x_nchw, gy, dw, db, stride, code, dy_nchw=False, y=None, act=ACT_NONE, accumulate=False):
    n, _, h, w = x_nchw.shape
    cout = dw.shape[0]
    if dy_nchw:
        ldg, yp, yld = 0, (y.data_ptr() if y is not None else None), 0
    else:
        ldg = nhwc_ld(gy)
        yp, yld = _pl(y)
    ws = workspace(_lib.load().wu_thin_workspace_bytes(), x_nchw.device)      # deterministic partial-sum slab
    _lib.call("wu_conv3x3_c3_wgrad", x_nchw.data_ptr(), gy.data_ptr(), ldg, 1 if dy_nchw else 0, yp, yld, act,
              dw.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), ws.numel(),
              n, h, w, cout, stride, 1 if accumulate else 0, code, stream_ptr())


def conv3x3_c3_dgrad(gy, weight, dx_nchw, stride, code, dy_nchw=False, y=None, act=ACT_NONE, accumulate=False):
    n, _, h, w = dx_nchw.shape
    cout = weight.shape[0]
    if dy_nchw:
        ldg, yp, yld = 0, (y.data_ptr() if y is not None else None), 0
    else:
        ldg = nhwc_ld(gy)
        yp, yld = _pl(y)
    _lib.call("wu_conv3x3_c3_dgrad", gy.data_ptr(), ldg, 1 if dy_nchw else 0, yp, yld, act, weight.data_ptr(), None,
              dx_nchw.data_ptr(), n, h, w, cout, stride, 1 if accumulate else 0, code, stream_ptr())


def conv3x3_relu_pool_bits(x, w_packed, bias, y, pool, gate_bits, sel_bits):
    """conv + bias + ReLU -> y, its 2x2 max-pool -> pool, and per element of y the ReLU-gate bit and the pool's arg-max bit
    (wu_conv3x3_relu_pool_bits_fwd; callers check gate_bits_supported(x, y) first)."""
    n, cin, h, w = x.shape
    cout = y.shape[1]
    _lib.call("wu_conv3x3_relu_pool_bits_fwd", x.data_ptr(), nhwc_ld(x), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), nhwc_ld(y), pool.data_ptr(), nhwc_ld(pool), gate_bits.data_ptr(), sel_bits.data_ptr(),
              n, h, w, cin, cout, dtype_code(x), stream_ptr())
    return y, pool


def conv3x3_small_supported(x, cout):
    """True when a stride-1 3x3 conv of x into `cout` channels belongs on the small-image kernel (wu_conv3x3_small_supported)."""
    if not x.is_cuda or dtype_code(x) != _lib.BF16:
        return False
    n, cin, h, w = x.shape
    return bool(_lib.load().wu_conv3x3_small_supported(n, h, w, nhwc_ld(x), cout, cout, cin, cout))


def chunk_major(w_packed):
    """[9][Cout][Cin] (pack_conv3x3) -> [Cin / 32][9][Cout][32]: the weight order of wu_conv3x3_small_fwd."""
    t, co, ci = w_packed.shape
    return w_packed.view(t, co, ci // 32, 32).permute(2, 0, 1, 3).contiguous()


def conv3x3_small(x, w_chunked, bias, y, act=ACT_NONE, egate=None, egate_act=ACT_NONE):
    """y = act(conv3x3(x) + bias) [* act'(egate)] on the small-image kernel with chunk-major weights (wu_conv3x3_small_fwd)."""
    n, cin, h, w = x.shape
    ep, eld = _pl(egate)
    _lib.call("wu_conv3x3_small_fwd", x.data_ptr(), nhwc_ld(x), w_chunked.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), nhwc_ld(y), ep, eld, egate_act, n, h, w, cin, y.shape[1], act, stream_ptr())
    return y


def conv3x3_head_supported(x, cout=64):
    """True when conv + ReLU + the 64 -> 3 head + tanh can run as ONE launch for input x (wu_conv3x3_relu_head_supported)."""
    if not x.is_cuda:
        return False
    n, cin, h, w = x.shape
    return bool(_lib.load().wu_conv3x3_relu_head_supported(h, w, nhwc_ld(x), cout, cin, cout, dtype_code(x)))


def conv3x3_relu_head(x, w_packed, bias, y, head_w, head_bias, out_nchw):
    """y = ReLU(conv3x3(x) + bias) and out_nchw = tanh(head_w . y + head_bias) in one launch (cunet.py:78-82); y may be None (the 64-channel
    tensor is then never written).  Callers check conv3x3_head_supported(x) first."""
    n, cin, h, w = x.shape
    _lib.call("wu_conv3x3_relu_head_fwd", x.data_ptr(), nhwc_ld(x), w_packed.data_ptr(), bias.data_ptr(),
              y.data_ptr() if y is not None else None, nhwc_ld(y) if y is not None else 64,
              head_w.data_ptr(), head_bias.data_ptr(), out_nchw.data_ptr(), n, h, w, cin, 64, dtype_code(x), stream_ptr())
    return y, out_nchw


def maxpool2_bwd_bits(gate_bits, sel_bits, gy, dx, dskip=None):
    """MaxPool2d(2) backward + skip-gradient sum + ReLU gate from the two bit planes the forward conv left (wu_maxpool2_bwd_bits)."""
    n, c, h, w = dx.shape
    sp, sld = _pl(dskip)
    _lib.call("wu_maxpool2_bwd_bits", gate_bits.data_ptr(), sel_bits.data_ptr(), gy.data_ptr(), nhwc_ld(gy), sp, sld, dx.data_ptr(), nhwc_ld(dx),
              n, h, w, c, dtype_code(dx), stream_ptr())
    return dx


def maxpool2(x, y):
    n, c, h, w = x.shape
    _lib.call("wu_maxpool2_fwd", x.data_ptr(), nhwc_ld(x), y.data_ptr(), nhwc_ld(y), n, h, w, c, dtype_code(x), stream_ptr())
    return y


def maxpool2_bwd(x, gy, dx, dskip=None, gate_act=ACT_NONE):
    n, c, h, w = x.shape
    sp, sld = _pl(dskip)
    _lib.call("wu_maxpool2_bwd", x.data_ptr(), nhwc_ld(x), gy.data_ptr(), nhwc_ld(gy), sp, sld, dx.data_ptr(), nhwc_ld(dx),
              n, h, w, c, gate_act, dtype_code(x), stream_ptr())
    return dx


def adain_stats(x, eps):
    """{mean, rstd} per (n, c): (N, C, 2) fp32 (utils.py:34-39,47)."""
    n, c, h, w = x.shape
    stats = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
    scratch = torch.empty((n, c, 2 * MAX_SPLITS), dtype=torch.float32, device=x.device)
    _lib.call("wu_adain_stats", x.data_ptr(), nhwc_ld(x), stats.data_ptr(), scratch.data_ptr(), n, h, w, c, float(eps),
              dtype_code(x), stream_ptr())
    return stats


def adain_upcat(x, stats, y_std, y_mean, cat, p_drop, seed, want_mask_bits, seed_dev=None, mask_in=None):
    """Writes channels [0, C) of `cat`; returns the keep-bit tensor (or None).
    ``seed_dev``: optional int64 device scalar added to `seed` inside the kernel (graph-safe per-replay masks);
    ``mask_in``: optional caller-supplied keep bits (``pack_keep_mask``) used INSTEAD of the counter RNG."""
    n, c, h, w = x.shape
    esz = x.element_size()
    nbits = n * 4 * h * w * (c * esz // 16)
    mbits = None
    if p_drop > 0 and mask_in is not None:
        if mask_in.dtype != torch.uint8 or mask_in.numel() != nbits or not mask_in.is_contiguous() or mask_in.device != x.device:
            raise ValueError(f"adain_upcat: mask_in must be {nbits} contiguous uint8 keep-bit bytes on {x.device}")
        mbits = mask_in
    elif p_drop > 0 and want_mask_bits:
        mbits = torch.empty(nbits, dtype=torch.uint8, device=x.device)
    _lib.call("wu_adain_upcat_fwd", x.data_ptr(), nhwc_ld(x), stats.data_ptr(), y_std.data_ptr(), y_mean.data_ptr(),
              cat.data_ptr(), nhwc_ld(cat), n, h, w, c, float(p_drop), int(seed),
              seed_dev.data_ptr() if seed_dev is not None else None,
              mbits.data_ptr() if mbits is not None else None, 1 if (mask_in is not None and p_drop > 0) else 0,
              dtype_code(x), stream_ptr())
    return mbits


def pack_keep_mask(mask_nchw, dtype):
    """(N, C, H2, W2) keep-mask (non-zero = keep) -> the kernels' keep-bit bytes: one byte per 16-byte channel chunk of the NHWC
    tensor, bit e = channel chunk*E + e (E = 8 bf16 / 4 fp32 elements)."""
    n, c, h2, w2 = mask_nchw.shape
    e = 16 // torch.empty((), dtype=dtype).element_size()
    m = (mask_nchw != 0).permute(0, 2, 3, 1).reshape(n, h2, w2, c // e, e).to(torch.int32)
    weights = (2 ** torch.arange(e, device=mask_nchw.device, dtype=torch.int32))
    return (m * weights).sum(-1).to(torch.uint8).reshape(-1).contiguous()


def adain_upcat_bwd(g_cat, x, stats, y_std, dx, p_drop, seed, mbits, x_gate_act=ACT_NONE):
    """Returns (d_y_std, d_y_mean); dx is written (optionally gated by act'(x))."""
    n, c, h, w = x.shape
    d_std = torch.empty((n, c), dtype=torch.float32, device=x.device)
    d_mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
    gtmp = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
    sums = torch.empty((n, c, 2 * (1 + MAX_SPLITS)), dtype=torch.float32, device=x.device)
    _lib.call("wu_adain_upcat_bwd", g_cat.data_ptr(), nhwc_ld(g_cat), x.data_ptr(), nhwc_ld(x), stats.data_ptr(), y_std.data_ptr(),
              dx.data_ptr(), nhwc_ld(dx), d_std.data_ptr(), d_mean.data_ptr(), gtmp.data_ptr(), sums.data_ptr(),
              n, h, w, c, float(p_drop), int(seed), mbits.data_ptr() if mbits is not None else None, x_gate_act,
              dtype_code(x), stream_ptr())
    return d_std, d_mean


def conv1x1_tanh(x, w3c, bias, out_nchw):
    n, cin, h, w = x.shape
    _lib.call("wu_conv1x1_tanh_fwd", x.data_ptr(), nhwc_ld(x), w3c.data_ptr(), bias.data_ptr(), out_nchw.data_ptr(),
              n, h, w, cin, dtype_code(x), stream_ptr())
    return out_nchw


def conv1x1_tanh_bwd(gout, out, x, w3c, dx, dw, db, x_gate_act=ACT_NONE, accumulate=False):
    n, cin, h, w = x.shape
    ws = workspace(_lib.load().wu_thin_workspace_bytes(), x.device)           # deterministic partial-sum slab
    _lib.call("wu_conv1x1_tanh_bwd", gout.data_ptr(), out.data_ptr(), x.data_ptr(), nhwc_ld(x), w3c.data_ptr(),
              dx.data_ptr(), nhwc_ld(dx), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
              n, h, w, cin, 1 if accumulate else 0, x_gate_act, dtype_code(x), stream_ptr())


def l1_mean(a, b, want_grad):
    """(mean|a - b| as a 0-dim fp32 tensor, sign(a - b) / n or None) for two contiguous fp32 tensors of the same shape."""
    n = a.numel()
    scratch = torch.empty(_lib.load().wu_l1_mean_scratch_floats(), dtype=torch.float32, device=a.device)
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    grad = torch.empty_like(a) if want_grad else None
    _lib.call("wu_l1_mean", a.data_ptr(), b.data_ptr(), grad.data_ptr() if want_grad else None, scratch.data_ptr(), loss.data_ptr(),
              n, stream_ptr())
    return loss, grad


def pack_conv3x3_multi(weights, code):
    """pack_conv3x3 for a list of OIHW fp32 weights in one launch (wu_pack_conv3x3_multi, <= 16 weights per launch)."""
    import ctypes
    from .layout import torch_dtype
    tdt = torch_dtype(code)
    outs, ws = [], []
    for weight in weights:
        cout, cin = weight.shape[:2]
        w = weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        ws.append(w)
        outs.append((torch.empty((9, cout, cin), dtype=tdt, device=weight.device), torch.empty((9, cin, cout), dtype=tdt, device=weight.device)))
    for i in range(0, len(ws), 16):
        chunk = list(range(i, min(i + 16, len(ws))))
        n = len(chunk)
        P = ctypes.c_void_p * n
        I = ctypes.c_int * n
        _lib.call("wu_pack_conv3x3_multi", n, P(*[ws[j].data_ptr() for j in chunk]), P(*[outs[j][0].data_ptr() for j in chunk]),
                  P(*[outs[j][1].data_ptr() for j in chunk]), I(*[ws[j].shape[0] for j in chunk]), I(*[ws[j].shape[1] for j in chunk]),
                  code, stream_ptr())
    return outs


def pack_conv3x3(weight, code):
    """(w_fwd [9][Cout][Cin], w_dgrad [9][Cin][Cout]) in the compute dtype."""
    from .layout import torch_dtype
    cout, cin = weight.shape[:2]
    tdt = torch_dtype(code)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    w_fwd = torch.empty((9, cout, cin), dtype=tdt, device=weight.device)
    w_dgrad = torch.empty((9, cin, cout), dtype=tdt, device=weight.device)
    _lib.call("wu_pack_conv3x3", w.data_ptr(), w_fwd.data_ptr(), w_dgrad.data_ptr(), cout, cin, None, code, stream_ptr())
    return w_fwd, w_dgrad
