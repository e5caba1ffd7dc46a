"""CPU oracle: stock-PyTorch fp32 restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Nothing under
``weather-unet_amd/`` imports this file; it is the checker, never the product.

Parity status: PINNED.  ``tests/golden/make_golden.py`` (run in the build
container, where ``/root/reference`` is importable) checks every function here
against the reference's own modules on the same weights/inputs (max-abs 0.0)
and writes the golden vectors in ``tests/golden/*.npz`` that
``tests/test_oracle_golden.py`` re-checks on every run.  The arithmetic lives
in PyTorch (reference pins torch==1.1.0, Pipfile:10; oracle of record is
torch 2.10 CPU fp32).

Each function cites the reference lines it follows.  All tensors are NCHW fp32
(the reference layout); parameters use the reference's state-dict key names.
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# deterministic parameter fill (build-owned; reproducible without the reference)
# --------------------------------------------------------------------------------------

G_CHANNELS = [(3, 64), (64, 128), (128, 256), (256, 512)]          # cunet.py:21-24
G_UP = [(256 + 512, 256), (128 + 256, 128), (64 + 128, 64)]        # cunet.py:34-36
ADAIN_CH = {"adain3": 512, "adain2": 256, "adain1": 128}           # cunet.py:30-32


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def _uniform(name, shape, bound, seed):
    a = _rng(name, seed).uniform(-bound, bound, size=shape).astype(np.float32)
    return torch.from_numpy(a)


def cunet_param_shapes(num_classes):
    """The 39 state-dict keys of Conditional_UNet (cunet.py:18-41) and their shapes."""
    shapes = {}
    blocks = {f"dconv_down{i + 1}": ch for i, ch in enumerate(G_CHANNELS)}
    blocks.update({f"dconv_up{3 - i}": ch for i, ch in enumerate(G_UP)})
    for name, (cin, cout) in blocks.items():
        shapes[f"{name}.0.weight"] = (cout, cin, 3, 3)       # nets.py:20
        shapes[f"{name}.0.bias"] = (cout,)
        shapes[f"{name}.2.weight"] = (cout, cout, 3, 3)      # nets.py:22
        shapes[f"{name}.2.bias"] = (cout,)
    for name, ch in ADAIN_CH.items():
        shapes[f"{name}.l1.weight"] = (4 * ch, num_classes)  # utils.py:31
        shapes[f"{name}.l1.bias"] = (4 * ch,)
        shapes[f"{name}.emb.weight"] = (num_classes, num_classes)  # utils.py:32 (unused)
    shapes["conv_last.weight"] = (3, 64, 1, 1)               # cunet.py:39
    shapes["conv_last.bias"] = (3,)
    return shapes


def make_cunet_params(num_classes=5, seed=0, gain=0.6):
    """Deterministic fill: conv weights U(-b, b), b = gain*sqrt(6/fan_in) (He-uniform x gain), small
    biases.  gain = 0.6 keeps every activation O(1) through the 15 convs while the pre-tanh output stays
    ~[-1, 1] (un-saturated tanh): a far harder parity case than nn.Conv2d's default init, under which the
    activations collapse towards the biases (that case is covered by the *_default_init goldens)."""
    params = {}
    for k, shp in sorted(cunet_param_shapes(num_classes).items()):
        if k.endswith("emb.weight"):
            params[k] = _uniform(k, shp, 1.0, seed)
        elif k.endswith("weight"):
            fan_in = int(np.prod(shp[1:]))
            g = gain * (math.sqrt(2.0) if ".l1." not in k and k != "conv_last.weight" else 1.0)
            params[k] = _uniform(k, shp, g * math.sqrt(3.0 / fan_in), seed)
        else:
            params[k] = _uniform(k, shp, 0.1, seed)
    return params


D_CHANNELS = [(3, 64), (64, 128), (128, 256), (256, 512)]          # disc.py:12-15


def sndisc_param_shapes(num_classes):
    """The 40 state-dict keys of SNDisc (disc.py:10-25): 10 spectral-norm layers x
    {bias, weight_orig, weight_u, weight_v}."""
    shapes = {}
    for i, (cin, cout) in enumerate(D_CHANNELS, start=1):
        for j, (ci, co) in enumerate([(cin, cin), (cin, cout)]):   # nets.py:28-31
            p = f"conv{i}.{j}"
            shapes[p + ".bias"] = (co,)
            shapes[p + ".weight_orig"] = (co, ci, 3, 3)
            shapes[p + ".weight_u"] = (co,)
            shapes[p + ".weight_v"] = (ci * 9,)
    for p, (co, ci) in {"l": (1, 512), "embed": (512, num_classes)}.items():  # disc.py:21,24
        shapes[p + ".bias"] = (co,)
        shapes[p + ".weight_orig"] = (co, ci)
        shapes[p + ".weight_u"] = (co,)
        shapes[p + ".weight_v"] = (ci,)
    return shapes


def make_sndisc_params(num_classes=5, seed=0):
    params = {}
    for k, shp in sorted(sndisc_param_shapes(num_classes).items()):
        if k.endswith("weight_orig"):
            fan_in = int(np.prod(shp[1:]))
            params[k] = _uniform(k, shp, math.sqrt(6.0 / fan_in), seed)
        elif k.endswith("weight_u") or k.endswith("weight_v"):
            v = torch.from_numpy(_rng(k, seed).standard_normal(shp).astype(np.float32))
            params[k] = F.normalize(v, dim=0, eps=1e-12)
        else:
            params[k] = _uniform(k, shp, 0.1, seed)
    return params


def make_inputs(batch, size, num_classes=5, seed=0, soft=False):
    """x ~ U(-1,1) (the Normalize(0.5,0.5) range, t_cls_train.py:93); c = one-hot rows
    (class = i % nc) or softmax(N(0,1)) rows."""
    r = _rng("inputs", seed)
    x = torch.from_numpy(r.uniform(-1, 1, size=(batch, 3, size, size)).astype(np.float32))
    if soft:
        c = torch.softmax(torch.from_numpy(r.standard_normal((batch, num_classes)).astype(np.float32)), 1)
    else:
        c = torch.eye(num_classes)[torch.arange(batch) % num_classes]
    return x, c


# --------------------------------------------------------------------------------------
# generator: Conditional_UNet.forward   (cunet.py:43-82)
# --------------------------------------------------------------------------------------

class _RoundBF16(torch.autograd.Function):
    """bf16-emulation hook (NOT part of the reference's arithmetic; see cunet_forward(emulate_bf16=True)): a tensor the
    MI355X kernels park in HBM as bf16.  Forward rounds the value to bf16 (round-to-nearest-even, kept in an fp32 tensor);
    backward rounds the gradient that flows into it -- the kernels store that gradient tensor in bf16 too."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


class _RoundGradBF16(torch.autograd.Function):
    """Identity forward, bf16-rounded gradient: a GRADIENT tensor the kernels park in bf16 where the forward keeps fp32."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _q(t, emu):
    return _RoundBF16.apply(t) if emu else t


def _qw(w, emu):
    """Weights are fp32 parameters re-packed to bf16 MFMA operands; their GRADIENT stays fp32 (straight-through)."""
    return w + (w.to(torch.bfloat16).float() - w).detach() if emu else w


def _conv3x3(x, w, b, order):
    """F.conv2d(padding=1); ``order`` (None or an int seed) evaluates the SAME sum with the input channels visited in a permuted
    order -- mathematically identical, different fp32 rounding: the summation-order witness of cunet_forward(emulate_bf16=True)."""
    if order is None:
        return F.conv2d(x, w, b, padding=1)
    perm = torch.randperm(x.shape[1], generator=torch.Generator().manual_seed(1000 * int(order) + x.shape[1]))
    return F.conv2d(x[:, perm].contiguous(), w[:, perm].contiguous(), b, padding=1)


def r_double_conv(p, name, x, emu=False, order=None):
    """nets.py:18-24: Conv3x3(pad 1, bias) -> ReLU -> Conv3x3 -> ReLU.
    (emu: bf16 operands, fp32 accumulate + bias + ReLU, bf16 store -- the kernels' storage points.)"""
    x = _q(F.relu(_conv3x3(x, _qw(p[f"{name}.0.weight"], emu), p[f"{name}.0.bias"], order)), emu)
    x = _q(F.relu(_conv3x3(x, _qw(p[f"{name}.2.weight"], emu), p[f"{name}.2.bias"], order)), emu)
    return x


def c_norm(x, bs, ch, eps):
    """utils.py:34-39: unbiased var over the last dim, +eps, sqrt; mean."""
    x_var = x.var(dim=-1) + eps
    x_std = x_var.sqrt().view(bs, ch, 1, 1)
    x_mean = x.mean(dim=-1).view(bs, ch, 1, 1)
    return x_std, x_mean


def adain(p, name, x, y, eps=1e-5):
    """utils.py:41-51.  y_ = l1(y).view(bs, ch, 4); both x and y_ normalised with eps=self.eps."""
    size = x.size()
    bs, ch = size[:2]
    x_ = x.view(bs, ch, -1)
    y_ = F.linear(y, p[f"{name}.l1.weight"], p[f"{name}.l1.bias"]).view(bs, ch, -1)
    x_std, x_mean = c_norm(x_, bs, ch, eps)
    y_std, y_mean = c_norm(y_, bs, ch, eps)
    return ((x - x_mean.expand(size)) / x_std.expand(size)) * y_std.expand(size) + y_mean.expand(size)


def upsample2(x):
    """cunet.py:26: nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def dropout(x, mask, p=0.3):
    """cunet.py:28 nn.Dropout(p=0.3): keep-mask (1 = keep) scaled by 1/(1-p); mask None = eval."""
    if mask is None:
        return x
    return x * mask.to(x.dtype) * (1.0 / (1.0 - p))


def cunet_forward(p, x, c, masks=None, return_stages=False, emulate_bf16=False, sum_order=None):
    """Conditional_UNet.forward (cunet.py:43-82).  ``masks`` = None (eval mode) or a list of
    three keep-masks for the dropouts at cunet.py:61,68,75 (train mode with a known mask).

    ``emulate_bf16`` (default False = the reference's fp32 arithmetic, the only mode pinned against the reference):
    the SAME graph with every tensor that the MI355X bf16 kernels keep in HBM as bf16 rounded to bf16 at that point --
    the input image and every conv weight as MFMA operands, each conv / pool / AdaIN-upsample-dropout output, and, through
    autograd, the gradient of each of those tensors (the gradient w.r.t. the AdaIN output is parked in bf16 as well).
    Accumulation, bias, activations, statistics, tanh and the weight gradients stay fp32, as in the kernels.  What is left
    between this mode and the HIP bf16 path is summation order (and the last-bit bf16 rounding flips it causes).

    ``sum_order`` (None or an int): visit every conv's input channels in a permuted order -- the same real-number function, a
    different fp32 rounding sequence.  Two emulated runs that differ ONLY in this are the yardstick for the bf16 gradient tests:
    measured (B=2, 64x64), their deep-layer weight gradients agree only to cosine 0.96-0.98 -- the instance-norm backward
    (utils.py:49-50 differentiated) removes the mean and x-hat components of a gradient that is dominated by exactly those
    components, so one-ulp bf16 flips in that gradient are amplified ~50x.  The HIP path is required to sit inside that
    emulation-to-emulation spread, which separates "bf16 precision mode" from "kernel error"."""
    emu = emulate_bf16
    order = sum_order
    m3, m2, m1 = masks if masks is not None else (None, None, None)
    st = {}
    x = _q(x, emu)                                           # (emu) the image enters the first conv's MFMA as bf16
    conv1 = r_double_conv(p, "dconv_down1", x, emu, order)          # :45
    x = _q(F.max_pool2d(conv1, 2), emu)                      # :46
    conv2 = r_double_conv(p, "dconv_down2", x, emu, order)          # :48
    x = _q(F.max_pool2d(conv2, 2), emu)                      # :49
    conv3 = r_double_conv(p, "dconv_down3", x, emu, order)          # :51
    x = _q(F.max_pool2d(conv3, 2), emu)                      # :52
    x = r_double_conv(p, "dconv_down4", x, emu, order)              # :54
    st.update(conv1=conv1, conv2=conv2, conv3=conv3, bottleneck=x)
    gq = _RoundGradBF16.apply if emu else (lambda t: t)
    x = gq(adain(p, "adain3", x, c))                         # :59
    st["adain3"] = x
    x = _q(torch.cat([dropout(upsample2(x), m3), conv3], dim=1), emu)  # :60-62
    x = r_double_conv(p, "dconv_up3", x, emu, order)                # :64
    st["up3"] = x
    x = gq(adain(p, "adain2", x, c))                         # :66
    st["adain2"] = x
    x = _q(torch.cat([dropout(upsample2(x), m2), conv2], dim=1), emu)  # :67-69
    x = r_double_conv(p, "dconv_up2", x, emu, order)                # :71
    st["up2"] = x
    x = gq(adain(p, "adain1", x, c))                         # :73
    st["adain1"] = x
    x = _q(torch.cat([dropout(upsample2(x), m1), conv1], dim=1), emu)  # :74-76
    x = r_double_conv(p, "dconv_up1", x, emu, order)                # :78
    st["up1"] = x
    out = torch.tanh(F.conv2d(x, p["conv_last.weight"], p["conv_last.bias"]))  # :80-82
    if return_stages:
        return out, st
    return out


# --------------------------------------------------------------------------------------
# discriminator: SNDisc.forward   (disc.py:27-38) with torch.nn.utils.spectral_norm
# --------------------------------------------------------------------------------------

def spectral_normalize(w_orig, u, v, train=True, eps=1e-12):
    """torch.nn.utils.spectral_norm (nets.py:28-31, disc.py:21,24), n_power_iterations=1.
    Train mode: v <- normalize(W^T u); u <- normalize(W v) (under no_grad, buffers updated),
    then sigma = u . (W v) and W / sigma.  Returns (w, u_new, v_new)."""
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(w_mat.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(w_mat, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w_orig / sigma, u, v


def _q_value(t, emu):
    """(emu) a tensor the kernels READ as bf16 but whose gradient they keep in fp32 (straight-through rounding)."""
    return t + (t.to(torch.bfloat16).float() - t).detach() if emu else t


def sn_double_conv(p, name, x, train, new_buffers, emu=False):
    """nets.py:26-33: SN(Conv3x3 cin->cin s1 p1) -> SN(Conv3x3 cin->cout **s2** p1) -> LeakyReLU(0.2);
    no activation between the two convs.

    ``emu`` (bf16-emulation, NOT part of the reference's arithmetic -- see cunet_forward): the storage points of the HIP bf16
    path.  The 3->3 conv of conv1 stays fp32 end to end (image layout, NCHW fp32 in and out); the 3->64 stride-2 conv reads that
    fp32 tensor and W/sigma as bf16 MFMA operands; every 64+-channel conv has bf16 operands, fp32 accumulate + bias
    (+ LeakyReLU), bf16 store.  Backward: the gradient of every bf16 tensor is parked in bf16, and the LeakyReLU-gated gradient
    is a second bf16 tensor (the kernels gate dY once, up front, and both gradient GEMMs read that)."""
    first = p[f"{name}.0.weight_orig"].shape[1] == 3
    gq = _RoundGradBF16.apply if emu else (lambda t: t)
    for j, stride in ((0, 1), (1, 2)):
        k = f"{name}.{j}"
        w, u, v = spectral_normalize(p[k + ".weight_orig"], p[k + ".weight_u"], p[k + ".weight_v"], train)
        new_buffers[k + ".weight_u"], new_buffers[k + ".weight_v"] = u, v
        if first and j == 0:
            x = F.conv2d(x, w, p[k + ".bias"], stride=stride, padding=1)                # fp32 kernel, fp32 NCHW output
            continue
        xin = _q_value(x, emu) if first else x
        z = F.conv2d(xin, _qw(w, emu), p[k + ".bias"], stride=stride, padding=1)
        x = _q(z, emu) if j == 0 else z
    return _q(F.leaky_relu(gq(x), 0.2), emu)


def sndisc_forward(p, x, c, train=True, emulate_bf16=False):
    """SNDisc.forward (disc.py:27-38).  Returns ([out, c1, c2, c3, c4], new_buffers) where
    new_buffers holds the power-iteration-updated weight_u / weight_v of the 10 SN layers.
    ``emulate_bf16``: see sn_double_conv (the pooled features and both linear heads stay fp32, as in the HIP path)."""
    nb = {}
    emu = emulate_bf16
    c1 = sn_double_conv(p, "conv1", x, train, nb, emu)       # :28
    c2 = sn_double_conv(p, "conv2", c1, train, nb, emu)      # :29
    c3 = sn_double_conv(p, "conv3", c2, train, nb, emu)      # :30
    c4 = sn_double_conv(p, "conv4", c3, train, nb, emu)      # :31
    feat = torch.sum(c4, [2, 3])                             # :32 global sum pool
    w, u, v = spectral_normalize(p["l.weight_orig"], p["l.weight_u"], p["l.weight_v"], train)
    nb["l.weight_u"], nb["l.weight_v"] = u, v
    out = F.linear(feat, w, p["l.bias"])                     # :33
    w, u, v = spectral_normalize(p["embed.weight_orig"], p["embed.weight_u"], p["embed.weight_v"], train)
    nb["embed.weight_u"], nb["embed.weight_v"] = u, v
    e_c = F.linear(c, w, p["embed.bias"])                    # :34
    out = out + torch.sum(e_c * feat, dim=1, keepdim=True)   # :36
    return [out, c1, c2, c3, c4], nb


# --------------------------------------------------------------------------------------
# losses (ops.py) -- restated for the step harness checks
# --------------------------------------------------------------------------------------

def dis_hinge(dis_fake, dis_real):
    """ops.py:42-45."""
    return torch.mean(torch.relu(1.0 - dis_real)) + torch.mean(torch.relu(1.0 + dis_fake))


def gen_hinge(dis_fake):
    """ops.py:47-48."""
    return torch.mean(-dis_fake)


def recon_loss(fake, img, pred, r):
    """t_cls_train.py:264-266: mean( mean|fake-img|_CHW / (mean|pred-r|_nc + 1e-7) )."""
    diff = torch.mean(torch.abs(fake - img), [1, 2, 3])
    lmda = torch.mean(torch.abs(pred - r), 1)
    return torch.mean(diff / (lmda + 1e-7))


def bench_loss(out, x):
    """The pure fwd+bwd benchmark's scalar loss: mean|G(x,c) - x|."""
    return torch.mean(torch.abs(out - x))


# --------------------------------------------------------------------------------------
# GAN step (t_cls_train.py:226-312, t_est_train.py:214-283) and evaluation() (t_cls_train.py:314-367)
# restated on the functional generator / discriminator above; `estimator` is any callable
# (N,3,H,W) -> (N,nc) RAW outputs (the pickled ResNet-101 of t_cls_train.py:172 is out of scope)
# --------------------------------------------------------------------------------------

def pred_loss(preds, labels, one_hot=False):
    """ops.py:29-40: CrossEntropy on raw outputs vs class indices if one_hot (--cross_ent) else MSE."""
    return F.cross_entropy(preds, labels) if one_hot else F.mse_loss(preds, labels)


def update_discriminator_loss(gp, dp, estimator_out, images, labels, c_d=None, supervised=False, masks=None):
    """t_cls_train.py:288-312 / t_est_train.py:261-283 up to d_loss.  ``estimator_out(x)`` is what the script calls
    ``self.estimator`` (softmax head in t_cls_train :174-178, the raw regressor in t_est_train).  Returns
    (d_loss, new_buffers_after_both_D_forwards)."""
    pred_labels = c_d if supervised else estimator_out(images).detach()                  # :294-297
    real, nb1 = sndisc_forward(dp, images, pred_labels, train=True)                     # :299
    with torch.no_grad():
        fake_out = cunet_forward(gp, images, labels, masks)                             # :302 (detached at :303)
    dp2 = dict(dp)
    dp2.update(nb1)
    fake, nb2 = sndisc_forward(dp2, fake_out, labels, train=True)                       # :303
    return dis_hinge(fake[0], real[0]), nb2                                             # :305


def update_inference_loss(gp, dp, estimator_out, estimator_raw, images, r_labels, d_labels=None, r_labels_=None,
                          supervised=False, cross_ent=False, masks=None):
    """t_cls_train.py:226-270 (flags --supervised / --cross_ent) and t_est_train.py:214-245 (both flags off).
    Returns (g_loss, g_loss_adv, loss_con, g_loss_w, g_loss_l1, fake_out)."""
    pred_labels = d_labels if supervised else estimator_out(images).detach()            # :233-237
    fake_out = cunet_forward(gp, images, r_labels, masks)                               # :242
    fake_d_out = sndisc_forward(dp, fake_out, r_labels, train=True)[0][0]               # :243-244
    if cross_ent:
        fake_c_out = estimator_raw(fake_out)                                            # :248 last layer is not softmax
    else:
        fake_c_out = estimator_out(fake_out)                                            # :250
        r_labels_ = r_labels                                                            # :251
    g_loss_adv = gen_hinge(fake_d_out)                                                  # :254
    g_loss_l1 = F.l1_loss(fake_out, images)                                             # :255 (logged only)
    g_loss_w = pred_loss(fake_c_out, r_labels_, one_hot=cross_ent)                      # :256
    diff = torch.mean(torch.abs(fake_out - images), [1, 2, 3])                          # :260,264
    lmda = torch.mean(torch.abs(pred_labels - r_labels), 1)                             # :261,265
    loss_con = torch.mean(diff / (lmda + (1e-2 if supervised else 1e-7)))               # :262,266
    g_loss = g_loss_adv + loss_con + g_loss_w                                           # :268-270 (lmda_con = lmda_w = 1)
    return g_loss, g_loss_adv, loss_con, g_loss_w, g_loss_l1, fake_out


def evaluation(gp, dp, estimator_out, estimator_eval, images, labels, ref_labels, d_train=False, masks=None):
    """t_cls_train.py:314-367 / t_est_train.py:285-332: for every reference row i, transfer the whole test batch to
    ref_labels[i] and average four losses over the B passes.  ``estimator_eval`` is ``self.estimator_`` (raw outputs) in
    t_cls_train (:338) and ``self.estimator`` in t_est_train (:309).  The scripts never call .eval(): G's dropout and D's
    power iteration stay active there.  ``d_train=True`` is that mode for D: each of the 2*B discriminator forwards
    (:340-341, under no_grad -- the spectral-norm hook iterates whenever ``module.training``, whatever the grad mode)
    advances weight_u / weight_v once, so every pass sees a slightly different W/sigma and D's buffers -- training state --
    have moved 2*B iterations when the sweep returns.  ``masks``: optional list of B dropout-mask triples for G's passes
    (train-mode G); None = eval-mode G (the deterministic part).
    Returns (dict of the four means, list of the B fake batches[, D's buffers after the sweep if d_train])."""
    bs, nc = images.shape[0], ref_labels.shape[1]
    adv, l1, w, d = [], [], [], []
    fakes = []
    dp = dict(dp)
    with torch.no_grad():
        for i in range(bs):
            ref_expand = torch.cat([ref_labels[i]] * bs).view(-1, nc)                   # :336
            fake = cunet_forward(gp, images, ref_expand, None if masks is None else masks[i])   # :337
            fake_c = estimator_eval(fake)                                               # :338
            real_o, nb = sndisc_forward(dp, images, labels, train=d_train)              # :340
            if d_train:
                dp.update(nb)
            fake_o, nb = sndisc_forward(dp, fake, ref_expand, train=d_train)            # :341
            if d_train:
                dp.update(nb)
            real_d, fake_d = real_o[0], fake_o[0]
            fakes.append(fake)
            adv.append(gen_hinge(fake_d).item())                                        # :349
            l1.append(F.l1_loss(fake, images).item())                                   # :350
            w.append(pred_loss(fake_c, ref_expand).item())                              # :351
            d.append(dis_hinge(fake_d, real_d).item())                                  # :352
    means = {"g_loss_adv": float(np.mean(adv)), "g_loss_l1": float(np.mean(l1)), "g_loss_w": float(np.mean(w)),
             "d_loss": float(np.mean(d))}
    if d_train:
        return means, fakes, {k: v for k, v in dp.items() if k.endswith(("weight_u", "weight_v"))}
    return means, fakes
