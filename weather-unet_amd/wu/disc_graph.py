"""The conv trunk of SNDisc (reference disc.py:27-32: four sn_double_conv blocks + the global sum pool) as ONE autograd node.

Why (round 4): after round 3 the GAN iteration's three discriminator forwards, its backward and the loss phases were HOST-bound
(profiles/r03_gan_phase_times.txt: enqueue time = wall time, 1.2 ms per D forward for ~0.5 ms of kernels).  Per forward the trunk was
nine ``autograd.Function.apply`` calls (8 convs + pool), each with its own ctx bookkeeping, and its backward nine node dispatches with an
activation-gate launch in front of each LeakyReLU layer.  Inside the trunk every producer / consumer pair is ours, so -- as
``wu/unet_graph.py`` does for the generator and ``wu/resnet.py`` for the estimator -- the whole thing runs as one node with a static
kernel schedule:

  * forward: the same eight conv launches and the pool, no per-layer autograd bookkeeping;
  * backward: the LeakyReLU gate of a block's output is applied in the epilogue of the data-gradient kernel that produces that
    gradient (the next block's stride-1 conv: ``egate`` / LeakyReLU) instead of a stand-alone pass -- the same two roundings in the
    same order, so every gradient is bit-identical to the per-layer path (tests/test_gpu_round4.py);
  * frozen weights (the generator update runs D with ``requires_grad_(False)``, wu/train_step.py) skip their weight-gradient kernels.

The spectral normalisation (one batched call, ``wu.functional.SpectralNormMultiFn``) and the two linear heads stay outside: their
gradients reach this node through the normalised weights it takes as inputs.
"""
import torch
from torch.autograd import Function

from . import kernels as K
from .layout import as_nhwc, empty_nhwc, precision_code, torch_dtype

import os

NONE, LEAKY = K.ACT_NONE, K.ACT_LEAKY
# the LeakyReLU gate of a block's output in the epilogue of the data-gradient conv that produces its gradient (True) or as a stand-alone
# in-place pass behind it (False); bit-identical either way (A/B switch WU_DISC_EGATE)
GATE_IN_EPILOGUE = os.environ.get("WU_DISC_EGATE", "1") == "1"


class SNDiscTrunkFn(Function):
    """(feat, c1, c2, c3, c4) = trunk(x; w0..w7, b0..b7).  ``meta`` = (code, packed): the compute dtype code and the six PackedConv
    caches of the wide convs (layers 2..7) with their identities; layers 0 / 1 (3 -> 3 stride 1 in the image layout, 3 -> 64 stride 2)
    read their OIHW fp32 weights directly."""

    @staticmethod
    def forward(ctx, meta, x, *wb):
        code, packed = meta
        # gradients of outputs the loss does not use arrive as None, NOT as materialised zero tensors: the GAN losses read `feat` only, and
        # zero-filled gradients of the four feature maps (67 + 34 + 17 + 8 MB at B = 32) followed by their adds cost 1 ms per iteration
        ctx.set_materialize_grads(False)
        dt, dev = torch_dtype(code), x.device
        ws, bs = wb[:8], wb[8:16]
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n, _, h, w = x.shape
        w01 = [ws[0].detach().contiguous(), ws[1].detach().contiguous()]
        t0 = K.conv3x3_c3(x, w01[0], bs[0], torch.empty((n, 3, h, w), dtype=torch.float32, device=dev), 1, NONE, True, code)
        hh, ww = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        c1 = K.conv3x3_c3(t0, w01[1], bs[1], empty_nhwc(n, 64, hh, ww, dt, dev), 2, LEAKY, False, code)
        feats, mids, wd = [c1], [t0], []
        cur = c1
        for blk in range(3):                                   # conv2, conv3, conv4: C -> C stride 1, C -> 2C stride 2 + LeakyReLU
            i = 2 + 2 * blk
            cin = cur.shape[1]
            pf0, pd0 = packed[i - 2][0].get(ws[i], code, packed[i - 2][1])
            pf1, pd1 = packed[i - 1][0].get(ws[i + 1], code, packed[i - 1][1])
            t = K.conv3x3(cur, pf0, bs[i], empty_nhwc(n, cin, hh, ww, dt, dev), 1, NONE)
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
            cur = K.conv3x3(t, pf1, bs[i + 1], empty_nhwc(n, 2 * cin, hh, ww, dt, dev), 2, LEAKY)
            mids.append(t)
            feats.append(cur)
            wd.extend((pd0, pd1))
        feat = torch.empty((n, cur.shape[1]), dtype=torch.float32, device=dev)
        K._lib.call("wu_sumpool_fwd", cur.data_ptr(), K.nhwc_ld(cur), feat.data_ptr(), n, hh, ww, cur.shape[1], code, K.stream_ptr())
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, *mids, *feats, *w01, *wd)
            ctx.code = code
            ctx.has_bias = [b is not None for b in bs]
            ctx.wshapes = [tuple(w_.shape) for w_ in ws]
        return (feat, *feats)

    @staticmethod
    def backward(ctx, g_feat, g_c1, g_c2, g_c3, g_c4):
        saved = ctx.saved_tensors
        x, mids, feats, w01, wd = saved[0], saved[1:5], saved[5:9], saved[9:11], saved[11:17]
        code = ctx.code
        dev, dt = x.device, feats[0].dtype
        n = x.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad                             # (meta, x, w0..w7, b0..b7)
        need_w = [need[2 + i] or (ctx.has_bias[i] and need[10 + i]) for i in range(8)]
        dws, dbs = [None] * 8, [None] * 8
        g_ext = [g_c1, g_c2, g_c3, g_c4]

        def wgrad(i, xin, gy, stride):
            if not need_w[i]:
                return
            dws[i] = torch.empty(ctx.wshapes[i], **f32)
            dbs[i] = torch.empty((ctx.wshapes[i][0],), **f32) if ctx.has_bias[i] else None
            K.conv3x3_wgrad(xin, gy, dws[i], dbs[i], stride)

        # gradient of the last block's output: the pool's broadcast (+ a gradient of the returned feature map), LeakyReLU-gated
        c4 = feats[3]
        _, c, hh, ww = c4.shape
        g = empty_nhwc(n, c, hh, ww, dt, dev)
        gf = g_feat.float().contiguous() if g_feat is not None else torch.zeros((n, c), **f32)
        K._lib.call("wu_sumpool_bwd", gf.data_ptr(), g.data_ptr(), K.nhwc_ld(g), n, hh, ww, c, code, K.stream_ptr())
        if g_ext[3] is not None:
            g = _add(g, as_nhwc(g_ext[3], code))
        g = K.act_gate(g, c4, LEAKY)
        for blk in (2, 1, 0):                                   # conv4, conv3, conv2
            i = 2 + 2 * blk
            t, cin_feat = mids[blk + 1], feats[blk]
            # stride-2 conv i+1: weight gradient, then its data gradient = gradient of the stride-1 conv's output (no activation between)
            wgrad(i + 1, t, g, 2)
            gt = K.conv3x3_s2_dgrad(g, wd[2 * blk + 1], empty_nhwc(*t.shape, dt, dev))
            wgrad(i, cin_feat, gt, 1)
            # stride-1 conv i: data gradient = gradient of the previous block's LeakyReLU output; its gate rides in the epilogue
            # (the stand-alone gate pass computes bf16(act'(y) * bf16(g)) -- so does the epilogue, on the packed result)
            gprev = empty_nhwc(*cin_feat.shape, dt, dev)
            if blk == 0:
                # c1's gate is applied INSIDE the first-layer kernels below (they take the ungated gradient + c1, as the per-layer path does)
                K.conv3x3(gt, wd[0], None, gprev)
                g = gprev if g_ext[0] is None else _add(gprev, as_nhwc(g_ext[0], code))
            elif g_ext[blk] is None and GATE_IN_EPILOGUE:
                K.conv3x3(gt, wd[2 * blk], None, gprev, egate=cin_feat, egate_act=LEAKY)
                g = gprev
            elif g_ext[blk] is None:
                K.conv3x3(gt, wd[2 * blk], None, gprev)
                g = K.act_gate(gprev, cin_feat, LEAKY, out=gprev)
            else:
                K.conv3x3(gt, wd[2 * blk], None, gprev)
                g = K.act_gate(_add(gprev, as_nhwc(g_ext[blk], code)), cin_feat, LEAKY)
        # conv1: 3 -> 64 stride 2 + LeakyReLU (gated inside the kernels from c1) after the image-layout 3 -> 3 conv
        t0, c1 = mids[0], feats[0]
        if need_w[1]:
            dws[1] = torch.empty(ctx.wshapes[1], **f32)
            dbs[1] = torch.empty((64,), **f32) if ctx.has_bias[1] else None
            K.conv3x3_c3_wgrad(t0, g, dws[1], dbs[1], 2, code, y=c1, act=LEAKY)
        dx = None
        if need[1] or need_w[0]:
            gt0 = torch.empty_like(t0)
            K.conv3x3_c3_dgrad(g, w01[1], gt0, 2, code, y=c1, act=LEAKY)
            if need_w[0]:
                dws[0] = torch.empty(ctx.wshapes[0], **f32)
                dbs[0] = torch.empty((3,), **f32) if ctx.has_bias[0] else None
                K.conv3x3_c3_wgrad(x, gt0, dws[0], dbs[0], 1, code, dy_nchw=True)
            if need[1]:
                dx = torch.empty_like(x)
                K.conv3x3_c3_dgrad(gt0, w01[0], dx, 1, code, dy_nchw=True)
        dws = [d if need[2 + i] else None for i, d in enumerate(dws)]
        dbs = [d if need[10 + i] else None for i, d in enumerate(dbs)]
        return (None, dx, *dws, *dbs)


def _add(a, b):
    """a + b for two NHWC-strided tensors of one shape (a gradient of a returned feature map joining the chain: off the GAN path)."""
    out = empty_nhwc(*a.shape, a.dtype, a.device)
    torch.add(a, b, out=out)
    return out


def trunk(x, weights, biases, packed, precision):
    """Run the conv trunk: ``weights`` = the eight (normalised) OIHW fp32 conv weights in forward order, ``biases`` their biases,
    ``packed`` = [(PackedConv, identity)] of layers 2..7.  Returns (feat (N, 512) fp32, c1, c2, c3, c4)."""
    code = precision_code(precision)
    return SNDiscTrunkFn.apply((code, packed), x, *weights, *biases)
