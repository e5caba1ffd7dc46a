"""CPU oracle for the estimator in the GAN loop: a stock-PyTorch fp32 restatement of torchvision's ResNet-101.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): nothing under ``weather-unet_amd/`` imports this file.

Parity status: **UNPINNED**.  The reference builds its classifier / estimator with ``torchvision.models.resnet101``
(classifier.py:106-112, estimator.py:143-151; Pipfile:11 pins ``torchvision<0.4``), a third-party dependency that is neither
vendored in /root/reference nor importable in this image (no network).  This file restates torchvision's published
architecture -- 7x7/2 stem, 3x3/2 max-pool, Bottleneck blocks [3, 4, 23, 3] with expansion 4 and the stride on the 3x3
conv, global average pool, ``fc`` -- with torchvision's state-dict key names so a checkpoint of the real model would load;
the reference ships no such checkpoint (its pickled models live at lab-local paths, t_cls_train.py:172) and holds no golden
vector for it, so nothing here could be checked against the real torchvision and the tests that use this file compare the
HIP estimator with THIS restatement only.

Use sites in the reference's GAN loop (the reason the estimator is on the hot path at all):
t_cls_train.py:237,247-250,297,424 -- four forwards per iteration and one data-gradient pass (g_loss flows through
``estimator(fake_out)`` into the generator); the estimator is frozen and in eval mode (:172-178).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = [(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)]      # (planes, blocks, stride) of resnet101
EXPANSION = 4
BN_EPS = 1e-5


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed, 101])


def resnet101_param_shapes(num_classes, layers=LAYERS):
    """torchvision ResNet state-dict keys (num_batches_tracked omitted: eval-mode forward never reads it)."""
    shapes = {"conv1.weight": (64, 3, 7, 7)}

    def bn(prefix, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{prefix}.{k}"] = (c,)
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(layers, start=1):
        for b in range(blocks):
            p = f"layer{li}.{b}"
            shapes[f"{p}.conv1.weight"] = (planes, inplanes, 1, 1)
            bn(f"{p}.bn1", planes)
            shapes[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{p}.bn2", planes)
            shapes[f"{p}.conv3.weight"] = (planes * EXPANSION, planes, 1, 1)
            bn(f"{p}.bn3", planes * EXPANSION)
            if b == 0 and (stride != 1 or inplanes != planes * EXPANSION):
                shapes[f"{p}.downsample.0.weight"] = (planes * EXPANSION, inplanes, 1, 1)
                bn(f"{p}.downsample.1", planes * EXPANSION)
            inplanes = planes * EXPANSION
    shapes["fc.weight"] = (num_classes, inplanes)
    shapes["fc.bias"] = (num_classes,)
    return shapes


def make_resnet101_params(num_classes=5, seed=0, layers=LAYERS):
    """Deterministic fill that keeps activations O(1) through 33 residual blocks: He-normal conv weights, BatchNorm statistics
    of a plausible trained model (running_var ~ U(0.5, 1.5), running_mean ~ N(0, 0.1), gamma ~ U(0.5, 1.0) -- 0.25 on each
    block's last BN so the residual sum does not grow), small betas."""
    p = {}
    for k, shp in sorted(resnet101_param_shapes(num_classes, layers).items()):
        r = _rng(k, seed)
        if k.endswith("running_var"):
            a = r.uniform(0.5, 1.5, size=shp)
        elif k.endswith("running_mean"):
            a = r.normal(0.0, 0.1, size=shp)
        elif ".bn" in k or k.startswith("bn1") or ".downsample.1" in k:
            if k.endswith("weight"):
                a = r.uniform(0.5, 1.0, size=shp) * (0.25 if ".bn3." in k else 1.0)
            else:
                a = r.normal(0.0, 0.05, size=shp)
        elif k == "fc.bias":
            a = r.normal(0.0, 0.05, size=shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = r.normal(0.0, math.sqrt(2.0 / fan_in), size=shp)
        p[k] = torch.from_numpy(a.astype(np.float32))
    return p


def _bn(p, prefix, x):
    """nn.BatchNorm2d in eval mode (the estimator is frozen, t_cls_train.py:173,178)."""
    return F.batch_norm(x, p[f"{prefix}.running_mean"], p[f"{prefix}.running_var"], p[f"{prefix}.weight"], p[f"{prefix}.bias"],
                        training=False, eps=BN_EPS)


def bottleneck(p, prefix, x, stride):
    """torchvision Bottleneck: 1x1 -> 3x3 (stride here) -> 1x1 (x4), BN after each, ReLU after the first two and after the sum."""
    out = F.relu(_bn(p, f"{prefix}.bn1", F.conv2d(x, p[f"{prefix}.conv1.weight"])))
    out = F.relu(_bn(p, f"{prefix}.bn2", F.conv2d(out, p[f"{prefix}.conv2.weight"], stride=stride, padding=1)))
    out = _bn(p, f"{prefix}.bn3", F.conv2d(out, p[f"{prefix}.conv3.weight"]))
    if f"{prefix}.downsample.0.weight" in p:
        x = _bn(p, f"{prefix}.downsample.1", F.conv2d(x, p[f"{prefix}.downsample.0.weight"], stride=stride))
    return F.relu(out + x)


def resnet101_forward(p, x, layers=LAYERS, return_stages=False):
    """torchvision ResNet.forward in eval mode -> (N, num_classes) RAW outputs (what the scripts call ``estimator_``)."""
    st = {}
    x = F.relu(_bn(p, "bn1", F.conv2d(x, p["conv1.weight"], stride=2, padding=3)))
    st["stem"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    st["pool"] = x
    for li, (planes, blocks, stride) in enumerate(layers, start=1):
        for b in range(blocks):
            x = bottleneck(p, f"layer{li}.{b}", x, stride if b == 0 else 1)
        st[f"layer{li}"] = x
    feat = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    st["feat"] = feat
    out = F.linear(feat, p["fc.weight"], p["fc.bias"])
    return (out, st) if return_stages else out


# ---------------------------------------------------------------------------------------------------------------------------
# bf16 emulation of the HIP estimator (round 4).  NOT torchvision's arithmetic: the same network with every tensor the MI355X
# kernels park in HBM as bf16 rounded at that point -- BatchNorm folded into the conv weight FIRST (wu/resnet.py: plan()), the
# folded weight rounded to bf16, the bias kept in fp32, each block's three activations and the downsample branch stored as bf16,
# and in backward each of the stage's gradient tensors (g_b, g_a, the downsample gradient, the block-input gradient) stored as
# bf16.  Used to pin the estimator's bf16 BACKWARD block by block (tests/test_gpu_round4.py), as cunet_forward(emulate_bf16=True)
# does for the generator.
# ---------------------------------------------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    """Value rounded to bf16 in forward (a tensor stored in bf16), the gradient flowing into it rounded in backward."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def fold_bn(p, conv, bn):
    """(weight, bias) of conv followed by eval-mode BatchNorm as ONE conv: what ResNet101Estimator.plan() computes."""
    scale = p[f"{bn}.weight"] / torch.sqrt(p[f"{bn}.running_var"] + BN_EPS)
    return p[f"{conv}.weight"] * scale.view(-1, 1, 1, 1), p[f"{bn}.bias"] - p[f"{bn}.running_mean"] * scale


def bottleneck_bf16(p, prefix, x, stride):
    """One Bottleneck in the HIP path's bf16 mode; `x` holds bf16-representable values.  Returns (pre, out): the fp32 sum before the
    final ReLU (the point the stage's upstream gradient enters: the kernels hand a block a gradient already gated by that ReLU) and the
    stored output."""
    q = _RoundBF16.apply
    rw = lambda w: w.to(torch.bfloat16).float()
    w1, b1 = fold_bn(p, f"{prefix}.conv1", f"{prefix}.bn1")
    w2, b2 = fold_bn(p, f"{prefix}.conv2", f"{prefix}.bn2")
    w3, b3 = fold_bn(p, f"{prefix}.conv3", f"{prefix}.bn3")
    a = q(F.relu(F.conv2d(x, rw(w1), b1)))
    b = q(F.relu(F.conv2d(a, rw(w2), b2, stride=stride, padding=1)))
    if f"{prefix}.downsample.0.weight" in p:
        wd, bd = fold_bn(p, f"{prefix}.downsample.0", f"{prefix}.downsample.1")
        idn = q(F.conv2d(x, rw(wd), bd, stride=stride))
    else:
        idn = x
    pre = F.conv2d(b, rw(w3), b3) + idn
    return pre, q(F.relu(pre))


def bottleneck_bf16_stage_grad(p, prefix, x, g_out_gated, stride):
    """The block-input gradient of one Bottleneck in bf16 mode, given the stage input `x` (a ReLU / max-pool output stored in bf16) and
    the upstream gradient ALREADY gated by the block's final ReLU (what the HIP stage receives): gated by x > 0 and rounded to bf16,
    as the HIP stage stores it for the previous block."""
    xin = x.detach().float().requires_grad_(True)
    pre, out = bottleneck_bf16(p, prefix, xin, stride)
    pre.backward(g_out_gated.float())
    g = xin.grad * (x > 0).float()
    return out.detach(), g.to(torch.bfloat16).float()


def resnet101_forward_bf16(p, x, layers=LAYERS):
    """The whole estimator in the HIP path's bf16 mode (see above): image and folded stem weights rounded to bf16 for the matrix-core
    stem, every stored activation bf16, pooled features and the classifier in fp32 -> (N, num_classes) raw outputs."""
    q = _RoundBF16.apply
    rw = lambda w: w.to(torch.bfloat16).float()
    ws, bs = fold_bn(p, "conv1", "bn1")
    h = q(F.relu(F.conv2d(q(x), rw(ws), bs, stride=2, padding=3)))
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for li, (planes, blocks, stride) in enumerate(layers, start=1):
        for b in range(blocks):
            h = bottleneck_bf16(p, f"layer{li}.{b}", h, stride if b == 0 else 1)[1]
    feat = torch.flatten(F.adaptive_avg_pool2d(h, 1), 1)
    return F.linear(feat, p["fc.weight"], p["fc.bias"])
