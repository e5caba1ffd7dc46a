"""torch.autograd.Function wrappers around libwu_kernels.so.

The reference defines no custom backward (everything is torch.autograd, t_cls_train.py:272,307);
here every hot op has a hand-written HIP forward and backward, glued into autograd so that
``loss.backward()`` / ``torch.optim.Adam`` keep working unchanged on the reference's parameters
(fp32, OIHW, same state-dict keys).
"""
import torch
from torch.autograd import Function

from . import _lib
from . import kernels as K
from .kernels import MAX_SPLITS, workspace  # noqa: F401  (re-exported: tests and callers size scratch with these)
from .layout import (as_nhwc, dtype_code, empty_nhwc, nhwc_ld, require_cuda, stream_ptr, torch_dtype)

ACT_NONE, ACT_RELU, ACT_LEAKY = _lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_LEAKY


def _grad_nhwc(g, code):
    """Incoming gradient -> NHWC storage of the compute dtype (no copy when it already is)."""
    return as_nhwc(g, code)


# ----------------------------------------------------------------------------------------------
# packed weights
# ----------------------------------------------------------------------------------------------
# A weight can change without its ``_version`` moving: ``torch.optim.Adam(fused=True)`` (and the other fused / capturable
# optimizers) update parameters through ``torch._fused_adam_``, which does NOT bump the version counter (measured: version 1 -> 1
# across steps, while the foreach / single-tensor paths go 1 -> 2 -> 3).  A cache keyed on (storage, version) alone would then keep
# handing the convs the weights of step 0.  Every optimizer step therefore bumps a process-wide generation that is part of the
# key: after ANY ``optimizer.step()`` every packed image is rebuilt on its next use (13 small launches per generator forward,
# batched into one by ``repack_stale``).  Updates made outside ``torch.optim`` with ops that skip the version counter must call
# ``invalidate_packed()``.
_WEIGHT_GENERATION = [0]
# bumped by explicit invalidate_packed() calls only (collective broadcasts, raw-pointer writes): what a FROZEN module's derived
# state (the estimator's folded BatchNorm, wu.resnet) keys on, so that optimizer steps of OTHER networks do not rebuild it
_EXTERNAL_GENERATION = [0]


def invalidate_packed():
    """Force every PackedConv / captured graph / folded estimator state to rebuild on its next use: weights were changed behind
    autograd's version counter (``dist.broadcast`` / ``dist.all_reduce`` on a parameter do NOT move ``_version`` -- measured on
    torch 2.10, gloo and nccl alike --, nor do raw-pointer writes)."""
    _WEIGHT_GENERATION[0] += 1
    _EXTERNAL_GENERATION[0] += 1


def _on_optimizer_step(optimizer, args, kwargs):
    _WEIGHT_GENERATION[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402
_STEP_HOOK = _register_step_hook(_on_optimizer_step)


class PackedConv:
    """[tap][Cout][Cin] (forward) and [tap'][Cin][Cout] (data-gradient) MFMA operand images of one
    OIHW fp32 weight, cached on (storage, version, optimizer-step generation, dtype) so a weight is repacked once per update.

    ``ident``: when `weight` is a TEMPORARY derived from a parameter (the spectral-norm W/sigma of nets.SNConv3x3, a fresh
    tensor per forward whose storage the caching allocator hands out again at the same point of the next iteration), its
    pointer / version say nothing about its contents: the owner then passes an explicit identity of what the temporary was
    computed from (parameter storage + version, buffer versions, power-iteration generation) and that is the key."""

    def __init__(self):
        self.key = None
        self.w_fwd = None
        self.w_dgrad = None

    @staticmethod
    def make_key(weight, code, ident=None):
        if ident is None:
            ident = (weight.data_ptr(), weight._version)
        return (ident, _WEIGHT_GENERATION[0], code, tuple(weight.shape))

    def stale(self, weight, code, ident=None):
        return self.make_key(weight, code, ident) != self.key

    def get(self, weight, code, ident=None):
        key = self.make_key(weight, code, ident)
        if key != self.key:
            self.w_fwd, self.w_dgrad = K.pack_conv3x3(weight, code)
            self.key = key
        return self.w_fwd, self.w_dgrad


def repack_stale(pairs, code):
    """Rebuild, in ONE launch, the packed images of every (PackedConv, weight) pair whose key is out of date."""
    todo = [(pc, w) for pc, w in pairs if pc.stale(w, code)]
    if not todo:
        return 0
    outs = K.pack_conv3x3_multi([w for _, w in todo], code)
    for (pc, w), (wf, wd) in zip(todo, outs):
        pc.w_fwd, pc.w_dgrad = wf, wd
        pc.key = pc.make_key(w, code)
    return len(todo)


# ----------------------------------------------------------------------------------------------
# conv3x3 (+bias +activation), MFMA implicit GEMM        nets.py:18-33
# ----------------------------------------------------------------------------------------------
class Conv3x3Fn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, packed, stride, act, out, ident=None):
        require_cuda(x, "conv3x3")
        code = dtype_code(x)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        w_fwd, w_dgrad = packed.get(weight, code, ident)
        if out is None:
            y = empty_nhwc(n, cout, ho, wo, x.dtype, x.device)
        else:
            assert tuple(out.shape) == (n, cout, ho, wo) and out.dtype == x.dtype
            y = out.detach()
        K.conv3x3(x, w_fwd, bias, y, stride, act)
        ctx.save_for_backward(x, y, w_dgrad)
        ctx.meta = (stride, act, code, tuple(weight.shape), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, w_dgrad = ctx.saved_tensors
        stride, act, code, wshape, has_bias = ctx.meta
        n, cin, h, w = x.shape
        cout = wshape[0]
        ho, wo = y.shape[2], y.shape[3]
        gy = _grad_nhwc(gy, code)
        if act != ACT_NONE:
            # activation backward once, up front: both gradient GEMMs then read a pre-gated dY (the gated
            # in-kernel paths of the C ABI stay available: mask / y arguments)
            gy = K.act_gate(gy, y, act)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(n, cin, h, w, x.dtype, x.device)
            if stride == 1:
                K.conv3x3(gy, w_dgrad, None, dx)
            else:
                K.conv3x3_s2_dgrad(gy, w_dgrad, dx)
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
            db = torch.empty((cout,), dtype=torch.float32, device=x.device) if has_bias else None
            K.conv3x3_wgrad(x, gy, dw, db, stride)
        return dx, dw, db, None, None, None, None, None


def conv3x3(x, weight, bias, packed, stride=1, act=ACT_NONE, out=None, ident=None):
    return Conv3x3Fn.apply(x, weight, bias, packed, stride, act, out, ident)


# ----------------------------------------------------------------------------------------------
# conv3x3 with 3 input channels, straight from the NCHW fp32 image     cunet.py:45, disc.py:28
# ----------------------------------------------------------------------------------------------
class ConvC3Fn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, act, out_nchw, code):
        require_cuda(x, "conv3x3_c3")
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n, cin, h, w = x.shape
        assert cin == 3
        cout = weight.shape[0]
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        wt = weight.detach().contiguous()
        if out_nchw:
            y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
        else:
            y = empty_nhwc(n, cout, ho, wo, torch_dtype(code), x.device)
        K.conv3x3_c3(x, wt, bias, y, stride, act, out_nchw, code)
        ctx.save_for_backward(x, y, wt)
        ctx.meta = (stride, act, code, out_nchw, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, wt = ctx.saved_tensors
        stride, act, code, out_nchw, has_bias = ctx.meta
        n, _, h, w = x.shape
        cout = wt.shape[0]
        gy = gy.float().contiguous() if out_nchw else _grad_nhwc(gy, code)
        gate = y if act != ACT_NONE else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):      # frozen in the generator update (wu/train_step.py)
            dw = torch.empty(wt.shape, dtype=torch.float32, device=x.device)
            db = torch.empty((cout,), dtype=torch.float32, device=x.device) if has_bias else None
            K.conv3x3_c3_wgrad(x, gy, dw, db, stride, code, dy_nchw=out_nchw, y=gate, act=act)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            K.conv3x3_c3_dgrad(gy, wt, dx, stride, code, dy_nchw=out_nchw, y=gate, act=act)
        return dx, dw, db, None, None, None, None


def conv3x3_c3(x, weight, bias, stride, act, out_nchw, code):
    return ConvC3Fn.apply(x, weight, bias, stride, act, out_nchw, code)


# ----------------------------------------------------------------------------------------------
# MaxPool2d(2)        cunet.py:27,46,49,52
# ----------------------------------------------------------------------------------------------
class MaxPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x, "maxpool2")
        code = dtype_code(x)
        n, c, h, w = x.shape
        if h % 2 or w % 2:
            raise ValueError(f"maxpool2: H and W must be even, got {h}x{w}")
        y = empty_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
        K.maxpool2(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        code = dtype_code(x)
        n, c, h, w = x.shape
        gy = _grad_nhwc(gy, code)
        dx = empty_nhwc(n, c, h, w, x.dtype, x.device)
        K.maxpool2_bwd(x, gy, dx)
        return dx


def maxpool2(x):
    return MaxPool2Fn.apply(x)


# ----------------------------------------------------------------------------------------------
# AdaIN statistics + fused apply / bilinear x2 / dropout / concat      utils.py:41-51, cunet.py:59-62
# ----------------------------------------------------------------------------------------------
def adain_stats(x, eps):
    """{mean, rstd} per (n, c): (N, C, 2) fp32."""
    return K.adain_stats(x, eps)


class AdaINUpCatFn(Function):
    """cat([dropout(upsample(adain(x, y))), skip], dim=1) with the skip tensor already resident in
    channels [C, C+Cs) of `catbuf` (its producer wrote it there): only channels [0, C) are written."""

    @staticmethod
    def forward(ctx, x, y_std, y_mean, skip, catbuf, eps, p_drop, seed, seed_dev=None, mask_in=None):
        require_cuda(x, "adain_upcat")
        code = dtype_code(x)
        n, c, h, w = x.shape
        cs = skip.shape[1]
        assert tuple(catbuf.shape) == (n, c + cs, 2 * h, 2 * w) and catbuf.dtype == x.dtype
        ld = nhwc_ld(catbuf)
        esz = catbuf.element_size()
        if skip.data_ptr() != catbuf.data_ptr() + c * esz or nhwc_ld(skip) != ld:
            catbuf[:, c:].copy_(skip)          # producer did not write in place: one copy, still no torch.cat
        stats = adain_stats(x, eps)
        ys = y_std.detach().float().contiguous()
        ym = y_mean.detach().float().contiguous()
        out = catbuf.detach()
        ctx.mbits = K.adain_upcat(x, stats, ys, ym, out, p_drop, seed, ctx.needs_input_grad[0], seed_dev, mask_in)
        ctx.save_for_backward(x, stats, ys)
        ctx.meta = (code, float(p_drop), int(seed), cs)
        return out

    @staticmethod
    def backward(ctx, g):
        x, stats, ys = ctx.saved_tensors
        code, p_drop, seed, cs = ctx.meta
        n, c, h, w = x.shape
        g = _grad_nhwc(g, code)
        dx = empty_nhwc(n, c, h, w, x.dtype, x.device)
        d_std, d_mean = K.adain_upcat_bwd(g, x, stats, ys, dx, p_drop, seed, ctx.mbits)
        dskip = g[:, c:] if ctx.needs_input_grad[3] else None     # a channel-slice view: no copy
        return dx, d_std, d_mean, dskip, None, None, None, None, None, None


def adain_upcat(x, y_std, y_mean, skip, catbuf, eps, p_drop, seed, seed_dev=None, mask_in=None):
    return AdaINUpCatFn.apply(x, y_std, y_mean, skip, catbuf, eps, p_drop, seed, seed_dev, mask_in)


def dropout_mask(n, c, h2, w2, p_drop, seed, device):
    """The keep-mask adain_upcat draws for (seed, p): (N, C, H2, W2) uint8 -- used by the parity tests to
    feed the oracle the very same mask."""
    m = torch.empty((n, c, h2, w2), dtype=torch.uint8, device=device)
    _lib.call("wu_dropout_mask", m.data_ptr(), n, h2, w2, c, float(p_drop), int(seed), stream_ptr())
    return m


# ----------------------------------------------------------------------------------------------
# conv_last (1x1) + tanh       cunet.py:39-40,80-82
# ----------------------------------------------------------------------------------------------
class Conv1x1TanhFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, "conv1x1_tanh")
        code = dtype_code(x)
        n, cin, h, w = x.shape
        wt = weight.detach().reshape(3, cin).contiguous()
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=x.device)
        K.conv1x1_tanh(x, wt, bias, out)
        ctx.save_for_backward(x, wt, out)
        ctx.wshape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, wt, out = ctx.saved_tensors
        code = dtype_code(x)
        n, cin, h, w = x.shape
        gout = gout.float().contiguous()
        dx = empty_nhwc(n, cin, h, w, x.dtype, x.device)
        dw = torch.empty((3, cin), dtype=torch.float32, device=x.device)
        db = torch.empty((3,), dtype=torch.float32, device=x.device)
        K.conv1x1_tanh_bwd(gout, out, x, wt, dx, dw, db)
        return dx, dw.view(ctx.wshape), db


def conv1x1_tanh(x, weight, bias):
    return Conv1x1TanhFn.apply(x, weight, bias)


# ----------------------------------------------------------------------------------------------
# discriminator head: global sum pool        disc.py:32
# ----------------------------------------------------------------------------------------------
class L1MeanFn(Function):
    """mean|a - b| (reference ops.py:22-24) with the gradient sign(a - b)/n produced by the forward's single pass."""

    @staticmethod
    def forward(ctx, a, b):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss, grad = K.l1_mean(a, b, need)
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad * gout
        return (g if ctx.needs_input_grad[0] else None), (-g if ctx.needs_input_grad[1] else None)


def l1_mean(a, b):
    """Fused mean absolute error for two same-shaped fp32 device tensors."""
    return L1MeanFn.apply(a.contiguous(), b.contiguous())


class SumPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x, "sumpool")
        code = dtype_code(x)
        n, c, h, w = x.shape
        feat = torch.empty((n, c), dtype=torch.float32, device=x.device)
        _lib.call("wu_sumpool_fwd", x.data_ptr(), nhwc_ld(x), feat.data_ptr(), n, h, w, c, code, stream_ptr())
        ctx.meta = (code, tuple(x.shape), x.dtype)
        return feat

    @staticmethod
    def backward(ctx, g):
        code, (n, c, h, w), dt = ctx.meta
        g = g.float().contiguous()
        dx = empty_nhwc(n, c, h, w, dt, g.device)
        _lib.call("wu_sumpool_bwd", g.data_ptr(), dx.data_ptr(), nhwc_ld(dx), n, h, w, c, code, stream_ptr())
        return dx


def sumpool(x):
    return SumPoolFn.apply(x)


# ----------------------------------------------------------------------------------------------
# layout boundary (differentiable): NCHW fp32 <-> NHWC compute dtype
# ----------------------------------------------------------------------------------------------
class ToNHWCFn(Function):
    @staticmethod
    def forward(ctx, x, code):
        ctx.src = (x.dtype,)
        return as_nhwc(x, code)

    @staticmethod
    def backward(ctx, g):
        from .layout import to_nchw_f32
        return to_nchw_f32(as_nhwc(g, dtype_code(g) if g.dtype in (torch.float32, torch.bfloat16) else _lib.F32)).to(ctx.src[0]), None


def to_nhwc(x, code):
    if x.dtype == torch_dtype(code):
        try:
            nhwc_ld(x)
            return x
        except ValueError:
            pass
    return ToNHWCFn.apply(x, code)


def adain_apply(x, y_std, y_mean, eps):
    """Stand-alone AdaIN (reference utils.py:41-51) for callers that use the module outside the fused
    U-Net path: torch ops on the GPU tensor (off the hot path; the U-Net uses adain_upcat)."""
    xf = x.float()
    n, c = xf.shape[:2]
    flat = xf.reshape(n, c, -1)
    x_std = (flat.var(dim=-1) + eps).sqrt().view(n, c, 1, 1)
    x_mean = flat.mean(dim=-1).view(n, c, 1, 1)
    out = (xf - x_mean) / x_std * y_std.view(n, c, 1, 1) + y_mean.view(n, c, 1, 1)
    return out.to(x.dtype).contiguous(memory_format=torch.channels_last)


# ----------------------------------------------------------------------------------------------
# spectral normalisation of a conv weight      nets.py:28-31 (torch.nn.utils.spectral_norm)
# ----------------------------------------------------------------------------------------------
class SpectralNormFn(Function):
    """W_eff = W_orig / sigma with one power iteration in training mode (u, v updated IN PLACE, like torch's
    hook).  4 kernel launches forward, 2 backward, instead of ~25 tiny torch kernels per layer and forward."""

    @staticmethod
    def forward(ctx, weight_orig, u, v, do_power_iteration, eps):
        require_cuda(weight_orig, "spectral_norm")
        w = weight_orig.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        rows = w.shape[0]
        cols = w.numel() // rows
        lib = _lib.load()
        scratch = torch.empty(lib.wu_spectral_norm_scratch_floats(rows, cols), dtype=torch.float32, device=w.device)
        sigma = torch.empty(2, dtype=torch.float32, device=w.device)
        w_eff = torch.empty_like(w)
        _lib.call("wu_spectral_norm_fwd", w.data_ptr(), rows, cols, u.data_ptr(), v.data_ptr(), 1 if do_power_iteration else 0,
                  float(eps), sigma.data_ptr(), w_eff.data_ptr(), scratch.data_ptr(), stream_ptr())
        # backward needs the u, v that defined sigma: clone (the buffers advance again on the next forward) -- but only when there
        # WILL be a backward: the generator update runs D with its parameters frozen (wu/train_step.py), evaluation() under no_grad;
        # two tiny device copies per SN layer and forward were 146 hipMemcpyAsync per GAN iteration
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(w, u.clone(), v.clone(), sigma)
        return w_eff

    @staticmethod
    def backward(ctx, g):
        w, u, v, sigma = ctx.saved_tensors
        g = g.float().contiguous()
        rows = w.shape[0]
        cols = w.numel() // rows
        dw = torch.empty_like(w)
        scratch = torch.empty(_lib.load().wu_spectral_norm_scratch_floats(rows, cols), dtype=torch.float32, device=w.device)
        _lib.call("wu_spectral_norm_bwd", g.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(),
                  dw.data_ptr(), rows, cols, scratch.data_ptr(), stream_ptr())
        return dw, None, None, None, None


def spectral_normalize(weight_orig, u, v, do_power_iteration, eps=1e-12):
    return SpectralNormFn.apply(weight_orig, u, v, do_power_iteration, eps)


_SN_SCRATCH = {}


def _sn_scratch_floats(rows, cols):
    k = (rows, cols)
    n = _SN_SCRATCH.get(k)
    if n is None:
        n = _SN_SCRATCH[k] = int(_lib.load().wu_spectral_norm_scratch_floats(rows, cols))
    return n


class SpectralNormMultiFn(Function):
    """SpectralNormFn for the n <= 16 weights of one network in ONE call (wu_spectral_norm_fwd_multi: 5 launches forward, 2 backward,
    whatever n): SNDisc normalises ten weights per forward (disc.py:11-24) and 50 + 20 launches of a few microseconds each were a
    quarter of its GPU time at B = 32.  Same per-weight arithmetic, bit-identical results.  One slab holds every scratch area,
    sigma pair and the copies of u / v the backward needs (written by the kernels: no clone launches)."""

    @staticmethod
    def forward(ctx, do_power_iteration, eps, n, *tensors):
        import ctypes
        ws, us, vs = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        require_cuda(ws[0], "spectral_norm")
        wd = []
        for w in ws:
            w = w.detach()
            wd.append(w if w.is_contiguous() else w.contiguous())
        rows = [w.shape[0] for w in wd]
        cols = [w.numel() // r for w, r in zip(wd, rows)]
        need_bwd = any(ctx.needs_input_grad[3:3 + n])
        # slab layout (floats): per weight [scratch | sigma(2, padded to 4) | u_save | v_save], every area 16-byte aligned
        off, areas = 0, []
        for r, c in zip(rows, cols):
            a = {"scratch": off}
            off += (_sn_scratch_floats(r, c) + 3) // 4 * 4
            a["sigma"] = off
            off += 4
            if need_bwd:
                a["u"] = off
                off += (r + 3) // 4 * 4
                a["v"] = off
                off += (c + 3) // 4 * 4
            areas.append(a)
        dev = wd[0].device
        slab = torch.empty(off, dtype=torch.float32, device=dev)
        base = slab.data_ptr()
        w_eff = [torch.empty_like(w) for w in wd]
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        at = lambda key: P(*[base + 4 * a[key] for a in areas])  # noqa: E731
        _lib.call("wu_spectral_norm_fwd_multi", n, P(*[w.data_ptr() for w in wd]), I(*rows), I(*cols), P(*[u.data_ptr() for u in us]),
                  P(*[v.data_ptr() for v in vs]), 1 if do_power_iteration else 0, float(eps), at("sigma"), P(*[w.data_ptr() for w in w_eff]),
                  at("scratch"), at("u") if need_bwd else None, at("v") if need_bwd else None, stream_ptr())
        if need_bwd:
            ctx.save_for_backward(slab, *wd)
            ctx.areas, ctx.dims = areas, (rows, cols)
        return tuple(w_eff)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes
        slab, *wd = ctx.saved_tensors
        rows, cols = ctx.dims
        n = len(wd)
        idx = [i for i in range(n) if gs[i] is not None and ctx.needs_input_grad[3 + i]]
        dws = [None] * n
        if idx:
            m = len(idx)
            g = [gs[i].float().contiguous() for i in idx]
            for i in idx:
                dws[i] = torch.empty_like(wd[i])
            base = slab.data_ptr()
            P, I = ctypes.c_void_p * m, ctypes.c_int * m
            at = lambda key: P(*[base + 4 * ctx.areas[i][key] for i in idx])  # noqa: E731
            # the forward's scratch areas are free again: they hold the <G, W> partials now
            _lib.call("wu_spectral_norm_bwd_multi", m, P(*[t.data_ptr() for t in g]), P(*[wd[i].data_ptr() for i in idx]), at("u"), at("v"),
                      at("sigma"), P(*[dws[i].data_ptr() for i in idx]), I(*[rows[i] for i in idx]), I(*[cols[i] for i in idx]),
                      at("scratch"), stream_ptr())
        return (None, None, None) + tuple(dws) + (None,) * (2 * n)


def spectral_normalize_multi(weights, us, vs, do_power_iteration, eps=1e-12):
    """[W_i / sigma_i] for lists of weight_orig / weight_u / weight_v (at most 16): see SpectralNormMultiFn."""
    n = len(weights)
    assert 0 < n <= 16 and len(us) == n and len(vs) == n
    return SpectralNormMultiFn.apply(do_power_iteration, eps, n, *weights, *us, *vs)


# ----------------------------------------------------------------------------------------------
# AdaIN style statistics: l1(y).view(N, C, 4) -> (std, mean)                    utils.py:41-48
# ----------------------------------------------------------------------------------------------
class AdaINStyleFn(Function):
    """(y_std, y_mean) from the conditioning vector in ONE kernel (and one for the backward into l1.weight / l1.bias) instead of
    the ~15 launch-sized stock kernels per decoder level and direction.  The conditioning input itself gets no gradient (labels)."""

    @staticmethod
    def forward(ctx, y, weight, bias, eps):
        n, nc = y.shape
        c = weight.shape[0] // 4
        y = y.contiguous()
        w = weight.detach().contiguous()
        y_std = torch.empty((n, c), dtype=torch.float32, device=y.device)
        y_mean = torch.empty_like(y_std)
        y4 = torch.empty((n, c, 4), dtype=torch.float32, device=y.device)
        _lib.call("wu_adain_style_fwd", y.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, float(eps),
                  y_std.data_ptr(), y_mean.data_ptr(), y4.data_ptr(), n, c, nc, stream_ptr())
        ctx.save_for_backward(y, y4, y_std, y_mean)
        ctx.meta = (tuple(weight.shape), bias is not None)
        return y_std, y_mean

    @staticmethod
    def backward(ctx, d_std, d_mean):
        y, y4, y_std, y_mean = ctx.saved_tensors
        wshape, has_bias = ctx.meta
        n, nc = y.shape
        c = wshape[0] // 4
        d_std = (d_std if d_std is not None else torch.zeros_like(y_std)).float().contiguous()
        d_mean = (d_mean if d_mean is not None else torch.zeros_like(y_mean)).float().contiguous()
        dw = torch.empty(wshape, dtype=torch.float32, device=y.device)
        db = torch.empty((wshape[0],), dtype=torch.float32, device=y.device) if has_bias else None
        _lib.call("wu_adain_style_bwd", d_std.data_ptr(), d_mean.data_ptr(), y.data_ptr(), y4.data_ptr(), y_std.data_ptr(), y_mean.data_ptr(),
                  dw.data_ptr(), db.data_ptr() if db is not None else None, n, c, nc, 0, stream_ptr())
        return None, dw, db, None


def adain_style(y, weight, bias, eps):
    return AdaINStyleFn.apply(y, weight, bias, eps)


class AdaINStyleMultiFn(Function):
    """AdaINStyleFn for several AdaIN layers that share the conditioning input -- the three decoder levels of a U-Net pass (cunet.py:59,66,73) --
    in ONE launch per direction (wu_adain_style_{fwd,bwd}_multi; per level the single call's arithmetic, bit-identical).
    apply(y, eps_tuple, w0, b0, w1, b1, ...) -> (std0, mean0, std1, mean1, ...)."""

    @staticmethod
    def forward(ctx, y, eps, *wb):
        import ctypes
        n, nc = y.shape
        y = y.contiguous()
        ws = [w.detach().contiguous() for w in wb[0::2]]
        bs = list(wb[1::2])
        L = len(ws)
        cs = [w.shape[0] // 4 for w in ws]
        f32 = dict(dtype=torch.float32, device=y.device)
        stds = [torch.empty((n, c), **f32) for c in cs]
        means = [torch.empty((n, c), **f32) for c in cs]
        y4s = [torch.empty((n, c, 4), **f32) for c in cs]
        P, FA, IA = ctypes.c_void_p * L, ctypes.c_float * L, ctypes.c_int * L
        _lib.call("wu_adain_style_fwd_multi", L, y.data_ptr(), P(*[w.data_ptr() for w in ws]), P(*[b.data_ptr() if b is not None else None for b in bs]),
                  FA(*[float(e) for e in eps]), P(*[t.data_ptr() for t in stds]), P(*[t.data_ptr() for t in means]), P(*[t.data_ptr() for t in y4s]),
                  n, IA(*cs), nc, stream_ptr())
        ctx.save_for_backward(y, *y4s, *stds, *means)
        ctx.meta = ([tuple(w.shape) for w in wb[0::2]], [b is not None for b in bs])
        out = []
        for s_, m_ in zip(stds, means):
            out.extend((s_, m_))
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        wshapes, has_bias = ctx.meta
        L = len(wshapes)
        saved = ctx.saved_tensors
        y, y4s, stds, means = saved[0], saved[1:1 + L], saved[1 + L:1 + 2 * L], saved[1 + 2 * L:1 + 3 * L]
        n, nc = y.shape
        f32 = dict(dtype=torch.float32, device=y.device)
        d_std = [(g if g is not None else torch.zeros_like(stds[i])).float().contiguous() for i, g in enumerate(grads[0::2])]
        d_mean = [(g if g is not None else torch.zeros_like(means[i])).float().contiguous() for i, g in enumerate(grads[1::2])]
        dws = [torch.empty(ws, **f32) for ws in wshapes]
        dbs = [torch.empty((ws[0],), **f32) if hb else None for ws, hb in zip(wshapes, has_bias)]
        P, IA = ctypes.c_void_p * L, ctypes.c_int * L
        _lib.call("wu_adain_style_bwd_multi", L, P(*[t.data_ptr() for t in d_std]), P(*[t.data_ptr() for t in d_mean]), y.data_ptr(),
                  P(*[t.data_ptr() for t in y4s]), P(*[t.data_ptr() for t in stds]), P(*[t.data_ptr() for t in means]),
                  P(*[t.data_ptr() for t in dws]), P(*[t.data_ptr() if t is not None else None for t in dbs]),
                  n, IA(*[ws[0] // 4 for ws in wshapes]), nc, 0, stream_ptr())
        out = [None, None]
        for dw, db in zip(dws, dbs):
            out.extend((dw, db))
        return tuple(out)


def adain_style_multi(y, layers):
    """layers: [(l1.weight, l1.bias, eps), ...] (<= 4) -> [(y_std, y_mean), ...] from one launch (and one in backward)."""
    flat = []
    for w, b, _ in layers:
        flat.extend((w, b))
    out = AdaINStyleMultiFn.apply(y, tuple(float(e) for _, _, e in layers), *flat)
    return [(out[2 * i], out[2 * i + 1]) for i in range(len(layers))]
