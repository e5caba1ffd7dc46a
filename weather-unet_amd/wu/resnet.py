"""The estimator of the reference's GAN loop -- a frozen, eval-mode torchvision ResNet-101 (classifier.py:106-112,
estimator.py:143-151; loaded as a pickle at t_cls_train.py:172-178) -- on this library's HIP kernels.

The reference calls it four times per iteration (``estimator(rand_images)`` :424, ``estimator(images)`` :297 and :237,
``estimator(fake_out)`` / ``estimator_(fake_out)`` :247-250) and differentiates the last call with respect to its INPUT:
``g_loss_w`` flows through the estimator into the generator (:256,270,272).  The weights never train (:173-178 ``.eval()``,
frozen parameters), so

  * every BatchNorm is a per-channel affine, folded once into the preceding conv's weight and bias;
  * only forward and DATA gradients exist: no weight gradients, no optimizer state;
  * the whole network is ONE autograd node with a static kernel schedule (as wu/unet_graph.py does for the generator): each
    data-gradient kernel hands its producer a gradient that is already gated by that producer's ReLU, the residual sums are
    epilogue adds, and under ``torch.no_grad()`` (three of the four calls) nothing is kept.

``ResNet101Estimator`` keeps torchvision's module tree NAMES (``conv1, bn1, layer1.0.conv1, ..., layer4.2.bn3, fc`` and the
``downsample.0 / downsample.1`` pairs) as plain parameter / buffer holders, so ``load_state_dict`` accepts a torchvision
ResNet-101 state-dict unchanged; torchvision itself is not needed.  The forward returns RAW outputs (what the scripts call
``estimator_``; ``t_cls_train`` wraps it in ``nn.Softmax`` to get ``estimator``).
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import kernels as K
from .layout import dtype_code, empty_nhwc, nhwc_ld, precision_code, require_cuda, stream_ptr, torch_dtype

RELU, NONE = K.ACT_RELU, K.ACT_NONE
# tracing: set to a list and the next backward appends, per Bottleneck (last block first), its stage input, stored output, the gated
# upstream gradient it received and the block-input gradient it produced; None = off (no cost, no references kept)
CAPTURE = None
LAYERS = ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))     # torchvision resnet101: (planes, blocks, stride)
EXPANSION = 4
BN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# launch helpers (C ABI: include/wu_kernels.h, "frozen ResNet-101 estimator")
# ----------------------------------------------------------------------------------------------
def conv1x1(x, w, bias, y, act=NONE, residual=None, egate=None, egate_act=NONE, in_stride=1, out_stride=1):
    """y = act(w . x + bias + residual) * act'(egate): rows = the coarse grid (see wu_conv1x1_fwd)."""
    n, cin, hin, win = x.shape
    _, cout, hout, wout = y.shape
    if in_stride == 2:
        hc, wc = (hin - 1) // 2 + 1, (win - 1) // 2 + 1
    elif out_stride == 2:
        hc, wc = hin, win
    else:
        hc, wc = hin, win
    rp, rld = (residual.data_ptr(), nhwc_ld(residual)) if residual is not None else (None, 0)
    ep, eld = (egate.data_ptr(), nhwc_ld(egate)) if egate is not None else (None, 0)
    _lib.call("wu_conv1x1_fwd", x.data_ptr(), nhwc_ld(x), w.data_ptr(), bias.data_ptr() if bias is not None else None, rp, rld,
              y.data_ptr(), nhwc_ld(y), n, hc, wc, in_stride, hin, win, out_stride, hout, wout, cin, cout, act,
              ep, eld, egate_act, dtype_code(x), stream_ptr())
    return y


# The pointwise convs with Cout % 128 == 0 and at least 128 tiles of 128 x 128 run on the persistent LDS-DMA kernel (conv1x1_pw3_kernel, round 4:
# bit-identical to the 64 x 64-tile kernel, 1.2-1.7x faster on the layer2-4 shapes).  WU_PW_PERSIST=0 puts every call back on the old kernel (A/B).
if os.environ.get("WU_PW_PERSIST", "1") == "0":
    _lib.call("wu_set_option", 15, 0)


# conv3 + residual + ReLU of a block and conv1 + ReLU of the next in ONE launch (and the mirror-image pair in backward).  OFF by default:
# measured (round 4, profiles/r04_chain_bench.txt, r04_gan_chain_ab.txt) the chained launch is bit-identical but not faster -- every
# 32-row workgroup streams both weight matrices (1 MB at layer3) from L2, 256 workgroups at once: 28 us against 30 us for the two launches
# at layer3, 1.3-1.6x slower at layers 1-2, GAN iteration +0.1 ms (B = 32) / +1.0 ms (B = 64).  WU_RESNET_CHAIN=1 switches it on.
CHAIN = os.environ.get("WU_RESNET_CHAIN", "0") == "1"


# Only where it pays (scratch/bench_chain.py, profiles/r04_chain_bench.txt): the chained launch keeps ONE 512-thread workgroup per CU and
# is bound by its own serial chain of memory round trips per 32-row tile -- at K1 = 256 (layer3: 8 k rows, 1 MB of weights per pair) it
# beats the two launches, at K1 = 64 / 128 (131 k / 33 k rows, weights of 64-256 KB) the two bandwidth-bound launches are 1.3-1.6x faster.
CHAIN_MIN_K1 = int(os.environ.get("WU_RESNET_CHAIN_MIN_K1", "256"))


def chain_supported(k1, c1, c2, code):
    return CHAIN and k1 >= CHAIN_MIN_K1 and bool(_lib.load().wu_conv1x1_chain_supported(int(k1), int(c1), int(c2), int(code)))


def frag_pack(w):
    """[Cout][K] -> MFMA-fragment order [Cout / 32][K / 16][k half][cout row][8]: the 64 lanes' 16-byte A fragments of one (cout block,
    K step) are 1 KiB contiguous (include/wu_kernels.h, wu_conv1x1_chain)."""
    co, k = w.shape
    return w.view(co // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def conv1x1_chain(x, wa, bias_a, res, act1, gate1, gate1_act, y1, wb, bias_b, act2, gate2, gate2_act, y2):
    """y1 = act1(wa . x + bias_a + res) * act'(gate1);  y2 = act2(wb . y1 + bias_b) * act'(gate2)  (wu_conv1x1_chain): all tensors on
    one unit-stride pixel grid; wa / wb in fragment order (``frag_pack``)."""
    n, k1, h, w = x.shape
    c1, c2 = y1.shape[1], y2.shape[1]
    ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    ld = lambda t: nhwc_ld(t) if t is not None else 0            # noqa: E731
    _lib.call("wu_conv1x1_chain", x.data_ptr(), nhwc_ld(x), wa.data_ptr(), ptr(bias_a), ptr(res), ld(res), act1, ptr(gate1), ld(gate1), gate1_act,
              y1.data_ptr(), nhwc_ld(y1), wb.data_ptr(), ptr(bias_b), act2, ptr(gate2), ld(gate2), gate2_act, y2.data_ptr(), nhwc_ld(y2),
              n * h * w, k1, c1, c2, dtype_code(x), stream_ptr())
    return y1, y2


def stem7x7(x_nchw, w, bias, y, act, code):
    n, _, h, w_ = x_nchw.shape
    _lib.call("wu_stem7x7_fwd", x_nchw.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), nhwc_ld(y),
              n, h, w_, act, code, stream_ptr())
    return y


def stem7x7_dgrad(gy, w, dx_nchw, code, accumulate=False):
    n, _, h, w_ = dx_nchw.shape
    _lib.call("wu_stem7x7_dgrad", gy.data_ptr(), nhwc_ld(gy), w.data_ptr(), dx_nchw.data_ptr(), n, h, w_, 1 if accumulate else 0, code, stream_ptr())
    return dx_nchw


def maxpool3s2(x, y, argmax=None):
    n, c, h, w = x.shape
    _lib.call("wu_maxpool3s2_fwd", x.data_ptr(), nhwc_ld(x), y.data_ptr(), nhwc_ld(y), argmax.data_ptr() if argmax is not None else None,
              n, h, w, c, dtype_code(x), stream_ptr())
    return y


def maxpool3s2_bwd(gy, argmax, x, dx, gate_act=NONE):
    n, c, h, w = dx.shape
    _lib.call("wu_maxpool3s2_bwd", gy.data_ptr(), nhwc_ld(gy), argmax.data_ptr(), x.data_ptr() if x is not None else None,
              nhwc_ld(x) if x is not None else 0, dx.data_ptr(), nhwc_ld(dx), n, h, w, c, gate_act, dtype_code(gy), stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------------
# parameter holders with torchvision's names
# ----------------------------------------------------------------------------------------------
class _ConvP(nn.Module):
    def __init__(self, cout, cin, k):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")      # torchvision ResNet.__init__
        self.weight = nn.Parameter(w, requires_grad=False)


class _BNP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def affine(self):
        """(scale, shift) of eval-mode nn.BatchNorm2d: y = x * scale + shift."""
        scale = self.weight / torch.sqrt(self.running_var + BN_EPS)
        return scale, self.bias - self.running_mean * scale


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _ConvP(planes, inplanes, 1), _BNP(planes)
        self.conv2, self.bn2 = _ConvP(planes, planes, 3), _BNP(planes)
        self.conv3, self.bn3 = _ConvP(planes * EXPANSION, planes, 1), _BNP(planes * EXPANSION)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(_ConvP(planes * EXPANSION, inplanes, 1), _BNP(planes * EXPANSION))
        else:
            self.downsample = None


class ResNet101Estimator(nn.Module):
    """Frozen eval-mode ResNet-101 -> (N, num_classes) raw outputs; input (N,3,H,W) fp32 NCHW with H, W >= 32."""

    def __init__(self, num_classes=5, precision="bf16", layers=LAYERS):
        super().__init__()
        self.layers_cfg = tuple(layers)
        self.conv1, self.bn1 = _ConvP(64, 3, 7), _BNP(64)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(layers, start=1):
            mods = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                mods.append(_Bottleneck(inplanes, planes, s, b == 0 and (s != 1 or inplanes != planes * EXPANSION)))
                inplanes = planes * EXPANSION
            setattr(self, f"layer{li}", nn.Sequential(*mods))
        self.fc = nn.Linear(inplanes, num_classes)
        for p in self.parameters():
            p.requires_grad_(False)
        precision_code(precision)
        self.precision = precision
        self._plan = None
        self._plan_key = None
        self.eval()

    def train(self, mode=True):
        # the reference keeps the estimator in eval mode for good (t_cls_train.py:173,178): BatchNorm folding relies on it
        return super().train(False)

    # ---- folded / packed operands, rebuilt when a tensor of the state changes ----
    def _state_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        # a TRAINABLE estimator may be updated by a fused optimizer, which does not move ``_version`` (wu.functional): then the
        # optimizer-step generation joins the key; the frozen estimator of the GAN loop keeps its plan across G / D steps
        # (explicit invalidate_packed() calls -- a collective broadcast moves no ``_version`` either -- reach both through
        # the external generation)
        from .functional import _EXTERNAL_GENERATION, _WEIGHT_GENERATION
        gen = _WEIGHT_GENERATION[0] if any(p.requires_grad for p in self.parameters()) else _EXTERNAL_GENERATION[0]
        return (self.precision, gen) + tuple((t.data_ptr(), t._version) for t in ts)

    def plan(self):
        key = self._state_key()
        if key == self._plan_key:
            return self._plan
        code = precision_code(self.precision)
        dt = torch_dtype(code)
        with torch.no_grad():
            def fold(conv, bn):
                scale, shift = bn.affine()
                return (conv.weight * scale.view(-1, 1, 1, 1)).float().contiguous(), shift.float().contiguous()

            def pw(conv, bn):
                w, b = fold(conv, bn)
                w2 = w.view(w.shape[0], w.shape[1])
                d = {"w": w2.to(dt).contiguous(), "wt": w2.t().to(dt).contiguous(), "b": b}
                if code == _lib.BF16 and w2.shape[0] % 32 == 0 and w2.shape[1] % 32 == 0:
                    d["wp"], d["wtp"] = frag_pack(d["w"]), frag_pack(d["wt"])          # MFMA-fragment order (wu_conv1x1_chain)
                return d

            def c3(conv, bn):
                w, b = fold(conv, bn)
                wf, wd = K.pack_conv3x3(w, code)
                return {"w": wf, "wd": wd, "b": b}
            ws, bs = fold(self.conv1, self.bn1)
            plan = {"stem_w": ws, "stem_b": bs, "blocks": []}
            for li in range(1, len(self.layers_cfg) + 1):
                for blk in getattr(self, f"layer{li}"):
                    plan["blocks"].append({"c1": pw(blk.conv1, blk.bn1), "c2": c3(blk.conv2, blk.bn2), "c3": pw(blk.conv3, blk.bn3),
                                           "ds": pw(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None,
                                           "stride": blk.stride, "planes": blk.conv1.weight.shape[0]})
        self._plan, self._plan_key = plan, key
        return plan

    def forward(self, x):
        require_cuda(x, "ResNet101Estimator")
        if x.shape[2] < 32 or x.shape[3] < 32:
            raise ValueError(f"ResNet101Estimator: input {tuple(x.shape)} is smaller than the network's stride (32)")
        feat = ResNetFn.apply(x, self.plan(), precision_code(self.precision))       # (N, 2048) fp32: global average pool
        return torch.nn.functional.linear(feat, self.fc.weight, self.fc.bias)


class GraphedEstimatorPass:
    """hipGraph replay of the frozen estimator's NO-GRAD pass over a static batch shape (t_cls_train.py:424,297,237: ``self.estimator(images)``
    under ``torch.no_grad()``): ~330 kernel launches per pass become one graph launch from the host (the GAN iteration's enqueue time was 56-81 %
    of its wall time, profiles/r04_gan_phase_times.txt; every launcher of the library is capture-safe, include/wu_kernels.h).  The same kernels
    in the same order on the same stream: results are bit-identical to the eager pass (tests/test_gpu_round4.py).

        g = GraphedEstimatorPass(est, (2 * B, 3, 256, 256))
        raw = g((rand_images, images))          # parts are copied into the static input one after the other along the batch

    The captured kernels read the folded / packed operands of ``est.plan()`` that existed at capture time: the pass keeps that plan alive and
    ``stale()`` says when the estimator's state has moved (load_state_dict, a broadcast): the caller captures again."""

    def __init__(self, est, shape, warmup=2):
        dev = next(est.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedEstimatorPass needs the estimator on a GPU")
        self.est, self.shape = est, tuple(shape)
        self.x = torch.zeros(self.shape, dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):         # first-use work (BatchNorm folding, chunk-major weights, kernel attributes) must not be captured
                est(self.x)
        cur.wait_stream(side)
        self.key = est._state_key()
        self._plan = est._plan              # the captured kernels hold raw pointers into these tensors
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = est(self.x)
        self.replays = 0

    def stale(self):
        return self.est._state_key() != self.key

    def __call__(self, parts):
        if isinstance(parts, torch.Tensor):
            parts = (parts,)
        o = 0
        for t in parts:
            if tuple(t.shape[1:]) != self.shape[1:]:
                raise ValueError(f"GraphedEstimatorPass: captured for {self.shape}, got a part of shape {tuple(t.shape)}")
            self.x[o:o + t.shape[0]].copy_(t)
            o += t.shape[0]
        if o != self.shape[0]:
            raise ValueError(f"GraphedEstimatorPass: captured for a batch of {self.shape[0]}, got {o} images")
        self.graph.replay()
        self.replays += 1
        return self.out.clone()             # the static output is overwritten by the next replay


class _GraphedResNetFn(Function):
    """One replay of a GraphedEstimatorGradPass as an autograd node: forward = copy + graph launch, backward = ResNetFn.backward (eager) on the
    activations the replay left in the graph's static buffers."""

    @staticmethod
    def forward(ctx, x, holder):
        holder.x.copy_(x)
        holder.graph.replay()
        holder.replays += 1
        ctx.plan, ctx.code, ctx.saved, ctx.stem, ctx.amax, ctx.xshape = holder.state
        ctx.holder, ctx.rid = holder, holder.replays
        return holder.feat.clone()

    @staticmethod
    def backward(ctx, gfeat):
        if ctx.holder.replays != ctx.rid:
            raise RuntimeError("GraphedEstimatorGradPass: the pass was replayed again before this backward ran -- its saved activations are "
                               "gone (one forward, then its backward; use the eager estimator for several live forwards)")
        return ResNetFn.backward(ctx, gfeat)[0], None


class GraphedEstimatorGradPass:
    """hipGraph replay of the frozen estimator's DIFFERENTIATED forward (t_cls_train.py:247-250: ``self.estimator_(fake_out)`` with the
    generator's graph attached) over a static batch shape.  The forward launches are captured with every activation the backward needs
    written into the graph's static buffers; a call copies the input in, replays, and returns raw outputs whose autograd node runs the usual
    eager backward on those buffers.  One live forward at a time: a second replay before the first one's backward raises in that backward.
    Bit-identical to the eager node (tests/test_gpu_round4.py)."""

    def __init__(self, est, shape, warmup=2):
        dev = next(est.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedEstimatorGradPass needs the estimator on a GPU")
        self.est, self.shape = est, tuple(shape)
        self.x = torch.zeros(self.shape, dtype=torch.float32, device=dev)
        code = precision_code(est.precision)
        plan = est.plan()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                ResNetFn.body(self.x, plan, code, True)
        cur.wait_stream(side)
        self.key = est._state_key()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.feat, self.state = ResNetFn.body(self.x, plan, code, True)
        self.replays = 0

    def stale(self):
        return self.est._state_key() != self.key

    def __call__(self, x):
        if tuple(x.shape) != self.shape:
            raise ValueError(f"GraphedEstimatorGradPass: captured for {self.shape}, got {tuple(x.shape)}")
        feat = _GraphedResNetFn.apply(x, self)
        return torch.nn.functional.linear(feat, self.est.fc.weight, self.est.fc.bias)


# ----------------------------------------------------------------------------------------------
# the network body as one autograd node
# ----------------------------------------------------------------------------------------------
def _chunked(c2, key):
    """Chunk-major copy of a frozen 3x3 conv's packed weights (K.chunk_major), made the first time the small-image kernel asks for it and kept
    with the plan entry (the plan is rebuilt when the estimator's state changes)."""
    ck = key + "_chunked"
    if ck not in c2:
        c2[ck] = K.chunk_major(c2[key])
    return c2[ck]


def _half(v, s):
    return (v - 1) // s + 1


class ResNetFn(Function):
    @staticmethod
    def forward(ctx, x, plan, code):
        feat, state = ResNetFn.body(x, plan, code, ctx.needs_input_grad[0])
        if state is not None:
            ctx.plan, ctx.code, ctx.saved, ctx.stem, ctx.amax, ctx.xshape = state
        return feat

    @staticmethod
    def body(x, plan, code, keep):
        """The forward launches; ``keep``: also return what backward needs (plan, code, per-block activations, stem, pool arg-max, input shape)."""
        dt, dev = torch_dtype(code), x.device
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n, _, h, w = x.shape

        def new(c, hh, ww):
            return empty_nhwc(n, c, hh, ww, dt, dev)
        h1, w1 = _half(h, 2), _half(w, 2)
        stem = stem7x7(x, plan["stem_w"], plan["stem_b"], new(64, h1, w1), RELU, code)        # conv1 + bn1 + relu
        h2, w2 = _half(h1, 2), _half(w1, 2)
        amax = torch.empty(n * h2 * w2 * 64, dtype=torch.uint8, device=dev) if keep else None
        cur = maxpool3s2(stem, new(64, h2, w2), amax)                                          # maxpool
        saved = []
        hh, ww = h2, w2
        blocks = plan["blocks"]
        a_next = None                       # conv1 + bn1 + relu of the coming block, when the previous block's chained launch produced it
        for bi, blk in enumerate(blocks):
            s, planes = blk["stride"], blk["planes"]
            a = a_next if a_next is not None else conv1x1(cur, blk["c1"]["w"], blk["c1"]["b"], new(planes, hh, ww), RELU)   # conv1 + bn1 + relu
            ho, wo = _half(hh, s), _half(ww, s)
            if s == 1 and K.conv3x3_small_supported(a, planes):                                 # layer3 / layer4: small-image kernel, chunk-major weights
                b = K.conv3x3_small(a, _chunked(blk["c2"], "w"), blk["c2"]["b"], new(planes, ho, wo), RELU)
            else:
                b = K.conv3x3(a, blk["c2"]["w"], blk["c2"]["b"], new(planes, ho, wo), s, RELU)  # conv2 (stride here) + bn2 + relu
            if blk["ds"] is not None:                                                           # downsample: conv1x1 stride s + bn
                idn = conv1x1(cur, blk["ds"]["w"], blk["ds"]["b"], new(planes * EXPANSION, ho, wo), NONE, in_stride=s)
            else:
                idn = cur
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            if nxt is not None and "wp" in blk["c3"] and "wp" in nxt["c1"] and chain_supported(planes, planes * EXPANSION, nxt["planes"], code):
                # conv3 + bn3 + add + relu, then the next block's conv1 + bn1 + relu on the tile that is still on chip
                out, a_next = conv1x1_chain(b, blk["c3"]["wp"], blk["c3"]["b"], idn, RELU, None, NONE, new(planes * EXPANSION, ho, wo),
                                            nxt["c1"]["wp"], nxt["c1"]["b"], RELU, None, NONE, new(nxt["planes"], ho, wo))
            else:
                out = conv1x1(b, blk["c3"]["w"], blk["c3"]["b"], new(planes * EXPANSION, ho, wo), RELU, residual=idn)   # conv3 + bn3 + add + relu
                a_next = None
            if keep:
                saved.append((cur, a, b, out))
            cur, hh, ww = out, ho, wo
        feat = torch.empty((n, cur.shape[1]), dtype=torch.float32, device=dev)
        _lib.call("wu_sumpool_fwd", cur.data_ptr(), nhwc_ld(cur), feat.data_ptr(), n, hh, ww, cur.shape[1], code, stream_ptr())
        feat.mul_(1.0 / (hh * ww))                                                              # adaptive_avg_pool2d(1)
        return feat, ((plan, code, saved, stem, amax, tuple(x.shape)) if keep else None)

    @staticmethod
    def backward(ctx, gfeat):
        plan, code, saved, stem, amax = ctx.plan, ctx.code, ctx.saved, ctx.stem, ctx.amax
        dt, dev = torch_dtype(code), gfeat.device
        n, _, h, w = ctx.xshape

        def new(c, hh, ww):
            return empty_nhwc(n, c, hh, ww, dt, dev)
        last = saved[-1][3]
        _, c, hh, ww = last.shape
        g = new(c, hh, ww)
        gf = (gfeat.float() * (1.0 / (hh * ww))).contiguous()
        _lib.call("wu_sumpool_bwd", gf.data_ptr(), g.data_ptr(), nhwc_ld(g), n, hh, ww, c, code, stream_ptr())
        g = K.act_gate(g, last, RELU, out=g)                        # gradient wrt the last block's pre-ReLU sum
        blocks = plan["blocks"]
        gb_pre = None                       # conv3^T of the block being entered, when the following block's chained launch produced it
        for bi in range(len(blocks) - 1, -1, -1):
            blk, (xin, a, b, out) = blocks[bi], saved[bi]
            s, planes = blk["stride"], blk["planes"]
            g_out = g
            # g: gradient of this block's output, already gated by its final ReLU
            gb = gb_pre if gb_pre is not None else conv1x1(g, blk["c3"]["wt"], None, new(planes, b.shape[2], b.shape[3]), NONE, egate=b, egate_act=RELU)
            ga = new(planes, a.shape[2], a.shape[3])
            if s == 1 and K.conv3x3_small_supported(gb, planes):
                K.conv3x3_small(gb, _chunked(blk["c2"], "wd"), None, ga, egate=a, egate_act=RELU)
            elif s == 1:
                K.conv3x3(gb, blk["c2"]["wd"], None, ga, egate=a, egate_act=RELU)
            else:
                K.conv3x3_s2_dgrad(gb, blk["c2"]["wd"], ga, egate=a, egate_act=RELU)
            if blk["ds"] is not None:
                skip = conv1x1(g, blk["ds"]["wt"], None, new(xin.shape[1], xin.shape[2], xin.shape[3]), NONE, out_stride=s)
            else:
                skip = g
            # gradient wrt the block input = conv1 path + identity path; gated by the ReLU that produced the input (every block
            # input is a ReLU output; for the first block it is max-pool(ReLU(stem)): x > 0 there implies the routed stem element > 0)
            prev = blocks[bi - 1] if bi > 0 else None
            g = new(xin.shape[1], xin.shape[2], xin.shape[3])
            if prev is not None and "wtp" in blk["c1"] and "wtp" in prev["c3"] and chain_supported(planes, xin.shape[1], prev["planes"], code):
                # ... and conv3^T of the previous block (gated by the ReLU of its 3x3 conv's output) from the tile that is still on chip
                pb = saved[bi - 1][2]
                gb_pre = new(prev["planes"], pb.shape[2], pb.shape[3])
                conv1x1_chain(ga, blk["c1"]["wtp"], None, skip, NONE, xin, RELU, g, prev["c3"]["wtp"], None, NONE, pb, RELU, gb_pre)
            else:
                conv1x1(ga, blk["c1"]["wt"], None, g, NONE, residual=skip, egate=xin, egate_act=RELU)
                gb_pre = None
            if CAPTURE is not None:      # tracing hook (tests: block-by-block gradient checks with the upstream gradient held fixed)
                CAPTURE.append({"block": bi, "stride": s, "xin": xin, "out": out, "g_out": g_out, "g_in": g})
        gstem = maxpool3s2_bwd(g, amax, stem, new(64, stem.shape[2], stem.shape[3]), gate_act=RELU)
        dx = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
        stem7x7_dgrad(gstem, plan["stem_w"], dx, code)
        return dx, None, None
