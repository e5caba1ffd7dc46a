"""The whole Conditional_UNet forward / backward (reference cunet.py:43-82 + autograd) as ONE autograd node
with a static kernel schedule.

Why a single node instead of one autograd.Function per layer (``wu.functional``, still used when the blocks
are called on their own): inside the network every producer / consumer pair is ours, so the schedule can
  * hand each conv a PRE-GATED output gradient -- the ReLU backward of layer L is applied in the epilogue of
    whichever kernel produces dL/d(out_L) (the next conv's data-gradient pass, the fused max-pool backward, the
    AdaIN/upsample backward, the 1x1 head backward), so no stand-alone activation-backward pass runs and both
    gradient GEMMs stage their operands with plain LDS-DMA;
  * sum the two gradients of every encoder skip tensor (max-pool path + concat path, cunet.py:46,62) inside
    the max-pool backward kernel instead of a separate elementwise add;
  * reuse the zero-copy concat buffers and never touch autograd's per-op bookkeeping (45 ops -> 1 node).
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from . import kernels as K
from .layout import empty_nhwc, precision_code, torch_dtype

RELU = K.ACT_RELU
# Weight-gradient kernels run on a second HIP stream: nothing in the backward chain consumes dW, so each one overlaps the
# data-gradient kernel that follows it and the tails / ramps of the persistent one-workgroup-per-CU launches fill each other.
SIDE_STREAM_WGRAD = True
# ReLU gates between a block's two convs travel as bits (1/16 of the tensor) where the LDS-DMA conv runs (A/B switch)
GATE_BITS = True
# the encoder blocks' second conv (conv + ReLU + fused 2x2 max-pool) also leaves the gate bit and the pool's arg-max bit of every output
# element, and the fused max-pool backward reads those 2 bits instead of the skip tensor itself (round 4; A/B switch)
POOL_BITS = os.environ.get("WU_POOL_BITS", "1") == "1"
# where the FIRST conv's weight gradient (the last kernel of the backward chain) is launched: True = side stream (round 2), False =
# main stream.  Round 3 trace: the side stream has no slack at the end of backward any more -- the last persistent weight gradient
# (down1.2), its reducer and this kernel ran one after the other for 230 us after the main stream's last data gradient had finished;
# on the main stream it runs beside that tail instead of behind it.
C3_WGRAD_ON_SIDE = os.environ.get("WU_C3_WGRAD_SIDE", "0") == "1"
# the 1x1 head + tanh computed in the epilogue of the last decoder conv (csrc/conv3x3_mfma_v2.hip, GATED = 5; cunet.py:78-82) instead of by a
# kernel of its own that re-reads the 64-channel tensor; an undifferentiated forward then never writes that tensor (round 4; A/B switch)
HEAD_FUSED = os.environ.get("WU_HEAD_FUSED", "1") == "1"
# the three AdaIN style MLPs of a pass in one launch per direction (wu_adain_style_{fwd,bwd}_multi); 0: one launch per layer (bit-identical)
STYLE_BATCHED = os.environ.get("WU_STYLE_BATCHED", "1") == "1"
# the decoder's dropout keep decisions travel to the backward as stored keep BYTES (1: the forward writes one byte per 16-byte chunk, the
# backward's LDS-ring kernel reads them) or are drawn AGAIN from the counter hash by the backward's marching kernel (0: nothing stored; same
# masks, bit-identical gradients).  A/B switch; a device-side seed offset (graph replay) or caller-supplied masks always store / read bytes.
KEEP_BITS_STORED = os.environ.get("WU_KEEP_BITS", "1") == "1"
_SIDE = {}
_ORDER_EVENTS = {}
LIGHT_EVENTS = os.environ.get("WU_LIGHT_EVENTS", "1") == "1"
SIDE_STREAM_LOG = []     # one entry per probe: which candidate won and the median timings (diagnostic)


def _side_stream(dev):
    """A second stream that really runs BESIDE the current one.  HIP multiplexes streams onto a few hardware queues (4 by
    default) round-robin: every fourth stream of the pool shares the current stream's queue and then serialises with it (measured:
    9.8 ms/step instead of 9.5, worse than one stream; which pool index collides depends on what else created streams first,
    e.g. an RCCL communicator).  So the stream is chosen by a one-off probe: a ~0.5 ms single-workgroup spin on the current stream
    and on each of five candidates, three rounds each -- the candidate with the shortest MEDIAN wall time (its spin overlaps the
    main stream's) wins; the choice and the timings are logged once (WU_LOG_SIDE_STREAM=1 prints them)."""
    main = torch.cuda.current_stream(dev)
    key = (dev, main.cuda_stream)
    s = _SIDE.get(key)
    if s is None:
        import os
        import statistics
        import time
        prio = int(os.environ.get("WU_SIDE_STREAM_PRIORITY", "0"))       # A/B switch: -1 = high priority (torch has two levels)
        cands = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(5)]
        spin = 1_000_000
        torch.cuda._sleep(spin)                      # warm the spin kernel
        times = [[] for _ in cands]
        for _ in range(3):
            for i, cand in enumerate(cands):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                torch.cuda._sleep(spin)
                with torch.cuda.stream(cand):
                    torch.cuda._sleep(spin)
                torch.cuda.synchronize(dev)
                times[i].append(time.perf_counter() - t0)
        med = [statistics.median(t) for t in times]
        best = min(range(len(cands)), key=lambda i: med[i])
        s = _SIDE[key] = cands[best]
        SIDE_STREAM_LOG.append({"device": str(dev), "chosen": best, "median_ms": [round(m * 1e3, 3) for m in med]})
        if os.environ.get("WU_LOG_SIDE_STREAM"):
            print(f"[wu] side stream for {dev}: candidate {best} of {len(cands)}, median spin-pair ms {SIDE_STREAM_LOG[-1]['median_ms']}",
                  flush=True)
    return s


def prepare_side_stream(dev=None):
    """Run the side-stream probe now (outside any timed region): called by bench.py / the training harness before warm-up."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
    return _side_stream(dev) if SIDE_STREAM_WGRAD else None


class _CudaStreamOps:
    """The three stream operations GradRouter needs, on real HIP streams (tests substitute recording fakes)."""

    def __init__(self, dev):
        self.dev = dev

    def current(self):
        return torch.cuda.current_stream(self.dev)

    def use(self, stream):
        return torch.cuda.stream(stream)

    def order_after(self, waiter, producer):
        """`waiter` runs what is enqueued on it from now on after everything enqueued on `producer` so far.  torch's
        ``waiter.wait_stream(producer)`` records a default-flag event -- a system-scope release on the producing stream, 14 times per
        step on the critical path; the two streams are on one device, so a timing-less event without that fence
        (wu_stream_order_after) orders them just as well.  WU_LIGHT_EVENTS=0 falls back to torch's (A/B switch)."""
        if not LIGHT_EVENTS:
            return waiter.wait_stream(producer)
        # one event per (thread, waiter, producer): two threads (autograd's device thread, a DDP hook, a second model on other streams)
        # interleaving record / wait on ONE shared event could order a waiter against the wrong stream's record (round 4)
        import threading
        key = (threading.get_ident(), self.dev, waiter.cuda_stream, producer.cuda_stream)
        ev = _ORDER_EVENTS.get(key)
        if ev is None:
            import ctypes
            out = ctypes.c_void_p()
            _lib.call("wu_event_create", ctypes.byref(out))
            ev = _ORDER_EVENTS[key] = out.value               # lives as long as the process (a handful per device and thread)
        _lib.call("wu_stream_order_after", waiter.cuda_stream, producer.cuda_stream, ev)


class GradRouter:
    """Where the fused backward's parameter gradients go, and in which stream order they are announced.

    Without a sink (plain autograd): fresh tensors, returned to autograd when the node finishes.  With a gradient sink
    (wu.ddp.GradBucketReducer.attach): the weight-gradient kernels accumulate straight into the parameters' bucket-view
    ``.grad`` and each layer is announced (``sink.grad_written``) the moment its kernel is enqueued, so a full bucket's
    all-reduce runs beside the rest of backward.  A bucket's collective is ordered after the stream that is current when
    its LAST gradient is announced, and a bucket mixes gradients produced on both streams (weight gradients on the side
    stream, thin-layer / head gradients on the main one): every announcement is therefore made on the side stream after
    it has caught up with the main one (side >= main >= every producer enqueued so far).  Host logic only -- the stream
    operations come in through ``ops`` so the ordering is testable without a GPU (tests/test_ddp_cpu.py)."""

    def __init__(self, sink, sink_params, shapes, alloc, ops, main, side):
        self.sink, self.SP, self.shapes, self.alloc, self.ops, self.main, self.side = sink, sink_params, shapes, alloc, ops, main, side
        self.grads = {}

    def bufs(self, iw):
        """(dw, db, accumulate) for parameters iw, iw+1: fresh tensors handed to autograd, or the bucket views."""
        if self.sink is None:
            return self.alloc(self.shapes[iw]), self.alloc(self.shapes[iw + 1]), False
        # first write after zero_grad() overwrites the zeroed view (no read-modify-write); later ones accumulate
        SP = self.SP
        return SP[iw].grad, SP[iw + 1].grad, not (self.sink.fresh(SP[iw]) and self.sink.fresh(SP[iw + 1]))

    def on_side(self, fn):
        """Run `fn` (kernel launches) on the side stream, ordered after everything enqueued on the main stream so far."""
        if self.side is None:
            return fn()
        self.ops.order_after(self.side, self.main)        # operands' producers are enqueued on the main stream
        with self.ops.use(self.side):
            return fn()

    def done(self, key, iw, dw, db):
        if self.sink is None:
            self.grads[key] = (dw, db)
            return
        self.grads[key] = (None, None)
        if self.side is not None and self.ops.current() != self.side:
            self.ops.order_after(self.side, self.main)
            with self.ops.use(self.side):
                self.sink.grad_written(self.SP[iw])
                self.sink.grad_written(self.SP[iw + 1])
        else:
            self.sink.grad_written(self.SP[iw])
            self.sink.grad_written(self.SP[iw + 1])


# tracing: set to a dict and the next fused backward leaves its stage inputs and every intermediate gradient tensor in it (the node
# is ONE autograd function, so autograd hooks cannot see inside); None = off (no cost, no references kept)
CAPTURE = None

BLOCKS = ("dconv_down1", "dconv_down2", "dconv_down3", "dconv_down4", "dconv_up3", "dconv_up2", "dconv_up1")


def _new(n, c, h, w, dt, dev):
    return empty_nhwc(n, c, h, w, dt, dev)


class UNetFn(Function):
    @staticmethod
    def forward(ctx, meta, x, ys3, ym3, ys2, ym2, ys1, ym1, *params):
        code, p_drop, seeds, eps, packed, sink, inj, seed_dev, enc_cache = meta
        inj = inj if inj is not None else (None, None, None)
        dt, dev = torch_dtype(code), x.device
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n, _, h, w = x.shape
        P = list(params)
        wb = {name: (P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) for i, name in enumerate(BLOCKS)}
        w_last, b_last = P[28], P[29]
        # after an optimizer step every packed operand image is out of date: rebuild them all in one launch
        from .functional import repack_stale
        stale = []
        for i, name in enumerate(BLOCKS):
            if name != "dconv_down1":
                stale.append((packed[2 * i], wb[name][0]))
            stale.append((packed[2 * i + 1], wb[name][2]))
        repack_stale(stale, code)
        pk = {}                                  # (w_fwd, w_dgrad) per MFMA conv
        for i, name in enumerate(BLOCKS):
            w0, _, w2, _ = wb[name]
            if name != "dconv_down1":
                pk[name + ".0"] = packed[2 * i].get(w0, code)
            pk[name + ".2"] = packed[2 * i + 1].get(w2, code)
        want_bits = any(ctx.needs_input_grad)
        # Encoder sharing (wu/train_step.py): the encoder (cunet.py:45-54) has no Dropout and does not see the condition, so two forwards
        # of the SAME input with the SAME weights -- the generator's no-grad pass of the discriminator update and its pass of the
        # generator update, t_cls_train.py:302,242 -- compute identical encoder activations.  With a cache dict from the caller the first
        # forward keeps them (with gate bits, whatever its grad mode: the second one differentiates through them) and the second reuses
        # them: its decoder overwrites the upsampled halves of the same concat buffers.  The caller vouches that input and weights are
        # unchanged in between; a shape / pointer mismatch recomputes.
        enc = None
        if enc_cache is not None:
            enc = enc_cache.get("state")
            if enc is not None and enc["key"] != (x.data_ptr(), x._version, tuple(x.shape), code):
                enc = None
        enc_bits = want_bits or enc_cache is not None

        if enc is None:
            cat1 = _new(n, 192, h, w, dt, dev)
            cat2 = _new(n, 384, h // 2, w // 2, dt, dev)
            cat3 = _new(n, 768, h // 4, w // 4, dt, dev)
        else:
            cat1, cat2, cat3 = enc["cat"]
        conv1, conv2, conv3 = cat1[:, 128:], cat2[:, 256:], cat3[:, 512:]

        # The ReLU gate of every block's first conv output ("mid") is needed once more, by the data-gradient pass of the block's
        # second conv: where the LDS-DMA conv runs, the forward leaves that gate as BITS (1/16 of the tensor) and the backward's
        # epilogue reads one dword per lane and row instead of the tensor (wu_kernels.h, "gate bits").
        gbits = {}

        def mid_conv(name, xin, out, bits=None):
            """First conv of block `name`: conv + bias + ReLU, with gate bits when wanted and supported."""
            if GATE_BITS and (want_bits if bits is None else bits) and K.gate_bits_supported(xin, out):
                gbits[name] = K.gate_bits_alloc(out)
                return K.conv3x3_bits(xin, pk[name + ".0"][0], wb[name][1], out, RELU, gate_bits_out=gbits[name])
            return K.conv3x3(xin, pk[name + ".0"][0], wb[name][1], out, 1, RELU)

        # ---- encoder (cunet.py:45-54) ----
        w_first = wb["dconv_down1"][0].detach().contiguous()
        if enc is None:
            a1 = _new(n, 64, h, w, dt, dev)
            if GATE_BITS and enc_bits and K.conv3x3_c3_bits_supported(x, w_first, wb["dconv_down1"][1], 1, code) and K.gate_bits_supported(a1, a1):
                gbits["dconv_down1"] = K.gate_bits_alloc(a1)
                K.conv3x3_c3_bits(x, w_first, wb["dconv_down1"][1], a1, gbits["dconv_down1"], 1, code)
            else:
                K.conv3x3_c3(x, w_first, wb["dconv_down1"][1], a1, 1, RELU, False, code)
            def pool_conv(name, xin, out, pooled):
                """Second conv of encoder block `name`: conv + bias + ReLU into the concat slice `out`, its 2x2 max-pool, and (when a
                backward will follow and the LDS-DMA conv runs) the gate / arg-max bits of `out` for the fused max-pool backward."""
                if POOL_BITS and GATE_BITS and enc_bits and K.gate_bits_supported(xin, out):
                    gbits[name + ".pool"] = (K.gate_bits_alloc(out), K.gate_bits_alloc(out))
                    return K.conv3x3_relu_pool_bits(xin, pk[name + ".2"][0], wb[name][3], out, pooled, *gbits[name + ".pool"])[1]
                return K.conv3x3_relu_pool(xin, pk[name + ".2"][0], wb[name][3], out, pooled)[1]

            p1 = pool_conv("dconv_down1", a1, conv1, _new(n, 64, h // 2, w // 2, dt, dev))
            a2 = mid_conv("dconv_down2", p1, _new(n, 128, h // 2, w // 2, dt, dev), enc_bits)
            p2 = pool_conv("dconv_down2", a2, conv2, _new(n, 128, h // 4, w // 4, dt, dev))
            a3 = mid_conv("dconv_down3", p2, _new(n, 256, h // 4, w // 4, dt, dev), enc_bits)
            p3 = pool_conv("dconv_down3", a3, conv3, _new(n, 256, h // 8, w // 8, dt, dev))
            a4 = mid_conv("dconv_down4", p3, _new(n, 512, h // 8, w // 8, dt, dev), enc_bits)
            b4 = K.conv3x3(a4, pk["dconv_down4.2"][0], wb["dconv_down4"][3], _new(n, 512, h // 8, w // 8, dt, dev), 1, RELU)
            if enc_cache is not None:
                enc_cache["state"] = {"key": (x.data_ptr(), x._version, tuple(x.shape), code), "cat": (cat1, cat2, cat3),
                                      "act": (a1, p1, a2, p2, a3, p3, a4, b4), "gbits": dict(gbits)}
                enc_cache["computed"] = enc_cache.get("computed", 0) + 1
        else:
            a1, p1, a2, p2, a3, p3, a4, b4 = enc["act"]
            gbits.update(enc["gbits"])
            enc_cache["reused"] = enc_cache.get("reused", 0) + 1

        # ---- decoder (cunet.py:59-78): adain -> upsample -> dropout -> cat fused, then r_double_conv ----
        ys = [t.detach().float().contiguous() for t in (ys3, ys2, ys1)]
        ym = [t.detach().float().contiguous() for t in (ym3, ym2, ym1)]
        st3 = K.adain_stats(b4, eps)
        keep_stored = want_bits and (KEEP_BITS_STORED or seed_dev is not None)
        mb3 = K.adain_upcat(b4, st3, ys[0], ym[0], cat3, p_drop, seeds[0], keep_stored, seed_dev, inj[0])
        u3a = mid_conv("dconv_up3", cat3, _new(n, 256, h // 4, w // 4, dt, dev))
        u3b = K.conv3x3(u3a, pk["dconv_up3.2"][0], wb["dconv_up3"][3], _new(n, 256, h // 4, w // 4, dt, dev), 1, RELU)
        st2 = K.adain_stats(u3b, eps)
        mb2 = K.adain_upcat(u3b, st2, ys[1], ym[1], cat2, p_drop, seeds[1], keep_stored, seed_dev, inj[1])
        u2a = mid_conv("dconv_up2", cat2, _new(n, 128, h // 2, w // 2, dt, dev))
        u2b = K.conv3x3(u2a, pk["dconv_up2.2"][0], wb["dconv_up2"][3], _new(n, 128, h // 2, w // 2, dt, dev), 1, RELU)
        st1 = K.adain_stats(u2b, eps)
        mb1 = K.adain_upcat(u2b, st1, ys[2], ym[2], cat1, p_drop, seeds[2], keep_stored, seed_dev, inj[2])
        u1a = mid_conv("dconv_up1", cat1, _new(n, 64, h, w, dt, dev))
        # ---- last decoder conv + head (cunet.py:78-82) ----
        w3c = w_last.detach().reshape(3, 64).contiguous()
        out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
        if HEAD_FUSED and w3c.data_ptr() % 16 == 0 and K.conv3x3_head_supported(u1a):
            # one launch: the head is 8 extra MFMAs per tile on the conv epilogue's registers; a forward nobody differentiates does not even
            # write the 64-channel tensor (it has no other consumer)
            u1b = _new(n, 64, h, w, dt, dev) if want_bits else None
            K.conv3x3_relu_head(u1a, pk["dconv_up1.2"][0], wb["dconv_up1"][3], u1b, w3c, b_last.detach(), out)
        else:
            u1b = K.conv3x3(u1a, pk["dconv_up1.2"][0], wb["dconv_up1"][3], _new(n, 64, h, w, dt, dev), 1, RELU)
            K.conv1x1_tanh(u1b, w3c, b_last, out)

        if want_bits:
            ctx.save_for_backward(x, a1, cat1, p1, a2, cat2, p2, a3, cat3, p3, a4, b4, u3a, u3b, u2a, u2b, u1a, u1b, out,
                                  st3, st2, st1, ys[0], ys[1], ys[2], w3c, wb["dconv_down1"][0].detach())
            ctx.pk_dgrad = {k: v[1] for k, v in pk.items()}
            ctx.mbits = (mb3, mb2, mb1)
            ctx.gbits = gbits
            ctx.meta = (code, p_drop, seeds, [tuple(p.shape) for p in P])
            # gradient sink (wu.ddp.GradBucketReducer.attach): weight gradients are accumulated straight into the parameters'
            # bucket-view .grad and announced layer by layer, so bucket all-reduces overlap the rest of this backward
            ctx.sink = sink if (sink is not None and sink.enabled and sink.owns(P)) else None
            ctx.sink_params = P if ctx.sink is not None else None
        return out

    @staticmethod
    def backward(ctx, gout):
        (x, a1, cat1, p1, a2, cat2, p2, a3, cat3, p3, a4, b4, u3a, u3b, u2a, u2b, u1a, u1b, out,
         st3, st2, st1, ys3, ys2, ys1, w3c, w_first) = ctx.saved_tensors
        code, p_drop, seeds, shapes = ctx.meta
        wd = ctx.pk_dgrad
        mb3, mb2, mb1 = ctx.mbits
        gbits = ctx.gbits
        dev, dt = x.device, u1b.dtype
        n, _, h, w = x.shape
        f32 = dict(dtype=torch.float32, device=dev)
        sink, SP = ctx.sink, ctx.sink_params
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if SIDE_STREAM_WGRAD else None
        router = GradRouter(sink, SP, shapes, lambda shp: torch.empty(shp, **f32), _CudaStreamOps(dev), main, side)
        grads, grad_bufs, done = router.grads, router.bufs, router.done
        keep = []          # operands of side-stream kernels stay referenced until the streams are joined

        def wgrad(name, j, xin, gy):
            iw = 4 * BLOCKS.index(name) + j
            dw, db, acc = grad_bufs(iw)
            router.on_side(lambda: (K.conv3x3_wgrad(xin, gy, dw, db, accumulate=acc), done((name, j), iw, dw, db)))
            if side is not None:
                keep.append((xin, gy, dw, db))

        def block_bwd(name, xin, mid, g_out_gated, need_dx=True):
            """r_double_conv backward given the PRE-GATED gradient of its output; returns dL/d(xin) (ungated)."""
            wgrad(name, 2, mid, g_out_gated)
            g_mid = _new(*mid.shape, dt, dev)
            if name in gbits and K.gate_bits_supported(g_out_gated, g_mid):
                K.conv3x3_bits(g_out_gated, wd[name + ".2"], None, g_mid, egate_bits=gbits[name])
            else:
                K.conv3x3(g_out_gated, wd[name + ".2"], None, g_mid, egate=mid, egate_act=RELU)
            if name == "dconv_down1":
                # the first conv's weight gradient is the LAST kernel of the backward chain and nothing consumes it (see C3_WGRAD_ON_SIDE)
                dw, db, acc = grad_bufs(0)
                if C3_WGRAD_ON_SIDE:
                    router.on_side(lambda: (K.conv3x3_c3_wgrad(xin, g_mid, dw, db, 1, code, accumulate=acc), done((name, 0), 0, dw, db)))
                    if side is not None:
                        keep.append((xin, g_mid, dw, db))
                else:
                    K.conv3x3_c3_wgrad(xin, g_mid, dw, db, 1, code, accumulate=acc)
                    done((name, 0), 0, dw, db)
                return g_mid
            wgrad(name, 0, xin, g_mid)
            if not need_dx:
                return None
            return K.conv3x3(g_mid, wd[name + ".0"], None, _new(*xin.shape, dt, dev))

        # head: dx gated by ReLU'(u1b)
        gout = gout.float().contiguous()
        g_u1b = _new(n, 64, h, w, dt, dev)
        dw_last, db_last, acc_last = grad_bufs(28)
        K.conv1x1_tanh_bwd(gout, out, u1b, w3c, g_u1b, dw_last.view(3, 64), db_last, x_gate_act=RELU, accumulate=acc_last)
        done(("conv_last", 0), 28, dw_last, db_last)

        # decoder level 1
        g_cat1 = block_bwd("dconv_up1", cat1, u1a, g_u1b)
        g_u2b = _new(*u2b.shape, dt, dev)
        dys1, dym1 = K.adain_upcat_bwd(g_cat1, u2b, st1, ys1, g_u2b, p_drop, seeds[2], mb1, x_gate_act=RELU)
        # decoder level 2
        g_cat2 = block_bwd("dconv_up2", cat2, u2a, g_u2b)
        g_u3b = _new(*u3b.shape, dt, dev)
        dys2, dym2 = K.adain_upcat_bwd(g_cat2, u3b, st2, ys2, g_u3b, p_drop, seeds[1], mb2, x_gate_act=RELU)
        # decoder level 3
        g_cat3 = block_bwd("dconv_up3", cat3, u3a, g_u3b)
        g_b4 = _new(*b4.shape, dt, dev)
        dys3, dym3 = K.adain_upcat_bwd(g_cat3, b4, st3, ys3, g_b4, p_drop, seeds[0], mb3, x_gate_act=RELU)
        # bottleneck + encoder: max-pool backward fused with the skip-gradient sum and the ReLU gate
        def pool_bwd(name, skip, g_pooled, g_skip):
            """max_pool2d backward + the skip tensor's second gradient (its concat slice) + its ReLU gate, in one pass: from the two bit
            planes the forward conv left, or from the skip tensor itself."""
            out = _new(*skip.shape, dt, dev)
            if name + ".pool" in gbits:
                return K.maxpool2_bwd_bits(*gbits[name + ".pool"], g_pooled, out, dskip=g_skip)
            return K.maxpool2_bwd(skip, g_pooled, out, dskip=g_skip, gate_act=RELU)

        g_p3 = block_bwd("dconv_down4", p3, a4, g_b4)
        g_conv3 = pool_bwd("dconv_down3", cat3[:, 512:], g_p3, g_cat3[:, 512:])
        g_p2 = block_bwd("dconv_down3", p2, a3, g_conv3)
        g_conv2 = pool_bwd("dconv_down2", cat2[:, 256:], g_p2, g_cat2[:, 256:])
        g_p1 = block_bwd("dconv_down2", p1, a2, g_conv2)
        g_conv1 = pool_bwd("dconv_down1", cat1[:, 128:], g_p1, g_cat1[:, 128:])
        g_a1 = block_bwd("dconv_down1", x, a1, g_conv1)
        dx = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty_like(x)
            K.conv3x3_c3_dgrad(g_a1, w_first.contiguous(), dx, 1, code)

        if side is not None:
            router.ops.order_after(main, side)
            keep.clear()
        if CAPTURE is not None:
            # tracing hook (tests/test_gpu_round3.py: stage-by-stage gradient checks with the upstream gradient held fixed)
            CAPTURE.update(x=x, cat1=cat1, cat2=cat2, cat3=cat3, p1=p1, p2=p2, p3=p3, b4=b4, u3b=u3b, u2b=u2b, u1b=u1b, out=out,
                           gout=gout, g_u1b=g_u1b, g_cat1=g_cat1, g_u2b=g_u2b, g_cat2=g_cat2, g_u3b=g_u3b, g_cat3=g_cat3, g_b4=g_b4,
                           g_p3=g_p3, g_conv3=g_conv3, g_p2=g_p2, g_conv2=g_conv2, g_p1=g_p1, g_conv1=g_conv1,
                           d_style=(dys3, dym3, dys2, dym2, dys1, dym1))
        flat = []
        for name in BLOCKS:
            for j in (0, 2):
                flat.extend(grads[(name, j)])
        gl = grads[("conv_last", 0)]
        flat.append(gl[0].view(shapes[28]) if gl[0] is not None else None)
        flat.append(gl[1])
        return (None, dx, dys3, dym3, dys2, dym2, dys1, dym1, *flat)


def unet_forward(net, x, c, encoder_cache=None):
    """Run ``Conditional_UNet`` `net` through the fused graph (called by its forward).  ``encoder_cache``: a dict shared by two forwards
    of the same input with unchanged weights -- the second reuses the first one's encoder activations (UNetFn.forward)."""
    code = precision_code(net.precision)
    c = c.to(device=x.device, dtype=torch.float32)
    styles = []
    ads = (net.adain3, net.adain2, net.adain1)
    if STYLE_BATCHED and c.is_cuda and not c.requires_grad and all(a.num_classes <= 32 for a in ads):
        # the three style MLPs (utils.py:41-48) in one launch per direction instead of three dependent launch-sized kernels at the top of
        # forward and at the very end of backward (bit-identical to the per-layer calls)
        from . import functional as WF
        for pair in WF.adain_style_multi(c, [(a.l1.weight, a.l1.bias, a.eps) for a in ads]):
            styles.extend(pair)
    else:
        for adain in ads:
            styles.extend(adain.style(c))
    params, packed = [], []
    for name in BLOCKS:
        blk = getattr(net, name)
        params.extend((blk[0].weight, blk[0].bias, blk[2].weight, blk[2].bias))
        packed.extend((blk[0]._packed, blk[2]._packed))
    params.extend((net.conv_last.weight, net.conv_last.bias))
    p = net.dropout.p if net.training else 0.0
    seeds = tuple(net._next_seed(k) for k in (3, 2, 1))
    inj = None
    if p > 0 and getattr(net, "dropout_masks", None) is not None:
        # caller-supplied keep-masks for the dropouts at cunet.py:61,68,75 (NCHW, non-zero = keep), e.g. masks captured from
        # the reference's nn.Dropout: the kernels read them instead of drawing from the counter RNG
        dt = torch_dtype(code)
        inj = tuple(K.pack_keep_mask(m.to(x.device), dt) for m in net.dropout_masks)
    meta = (code, float(p), seeds, float(net.adain3.eps), packed, getattr(net, "grad_sink", None), inj,
            getattr(net, "_seed_dev", None), encoder_cache)
    return UNetFn.apply(meta, x, *styles, *params)
