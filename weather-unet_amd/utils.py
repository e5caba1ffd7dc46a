"""Reference utils.py names.  ``AdaIN`` (reference utils.py:26-51) is on the hot path; the others are
imported by the reference's scripts (cunet.py:3, disc.py:5, t_cls_train.py:61) but never instantiated
there -- they are kept importable as small stock-torch modules."""
import torch
import torch.nn as nn

from wu import functional as WF
from wu.layout import precision_code, require_cuda


class AdaIN(nn.Module):
    """Condition injection (reference utils.py:26-51).

    ``y_ = l1(y).view(bs, ch, 4)``: per channel the mean / sqrt(unbiased var + eps) of those four numbers
    are the target statistics; x is instance-normalised over H*W (unbiased var, eps inside the sqrt) and
    re-scaled.  ``emb`` is unused by forward (reference utils.py:32) but lives in the state-dict.

    ``style(y)`` gives (y_std, y_mean); ``Conditional_UNet`` feeds them to the fused
    AdaIN -> bilinear x2 -> dropout -> concat kernel.  ``forward(x, y)`` is the stand-alone module call.
    """

    def __init__(self, in_channel, num_classes, eps=1e-5):
        super().__init__()
        self.num_classes = num_classes
        self.in_channel = in_channel
        self.eps = eps
        self.l1 = nn.Linear(num_classes, in_channel * 4, bias=True)
        self.emb = nn.Embedding(num_classes, num_classes)
        self.precision = "bf16"

    def style(self, y):
        """(y_std, y_mean), each (N, C) fp32 -- reference utils.py:46,48 (c_norm with eps=self.eps)."""
        bs = y.size(0)
        if y.is_cuda and not y.requires_grad and self.num_classes <= 32:
            return WF.adain_style(y.float(), self.l1.weight, self.l1.bias, self.eps)      # one fused kernel (+ one backward)
        y_ = self.l1(y.float()).view(bs, self.in_channel, -1)
        y_std = (y_.var(dim=-1) + self.eps).sqrt()
        y_mean = y_.mean(dim=-1)
        return y_std, y_mean

    def forward(self, x, y):
        require_cuda(x, "AdaIN")
        assert x.size(0) == y.size(0)
        code = precision_code(self.precision)
        x = WF.to_nhwc(x, code)
        y_std, y_mean = self.style(y)
        return WF.adain_apply(x, y_std, y_mean, self.eps)


class ConditionalNorm(nn.Module):
    """reference utils.py:7-23 (imported by disc.py:5, never instantiated)."""

    def __init__(self, in_channel, num_classes=5):
        super().__init__()
        self.num_classes = num_classes
        self.bn = nn.BatchNorm2d(in_channel, affine=False)
        self.embed = nn.Embedding(num_classes, in_channel * 2)
        self.embed.weight.data[:, :in_channel] = 1
        self.embed.weight.data[:, in_channel:] = 0

    def forward(self, input, class_id):
        out = self.bn(input)
        gamma, beta = self.embed(class_id).chunk(2, 1)
        return gamma.unsqueeze(2).unsqueeze(3) * out + beta.unsqueeze(2).unsqueeze(3)


class BatchNorm(nn.Module):
    """reference utils.py:54-71: per-sample whitening over C*H*W (unused on the executed path)."""

    def forward(self, x):
        bs = x.size(0)
        x_ = x.reshape(bs, -1)
        std = (x_.var(dim=-1) + 1e-5).sqrt().view(bs, 1, 1, 1)
        mean = x_.mean(dim=-1).view(bs, 1, 1, 1)
        return (x - mean) / std


class MakeOneHot(nn.Module):
    """reference utils.py:74-81."""

    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes

    def forward(self, x):
        return nn.functional.one_hot(torch.argmax(x), self.num_classes)


class HalfDropout(nn.Module):
    """reference utils.py:84-95: dropout on the first half of the channels."""

    def __init__(self, p=0.3):
        super().__init__()
        self.dropout = nn.Dropout(p=p)

    def forward(self, x):
        ch = x.size(1)
        return torch.cat([self.dropout(x[:, :ch // 2]), x[:, ch // 2:]], dim=1)


class Denormalize(object):
    """reference utils.py:98-109 (torchvision-free restatement of F.normalize with the inverse stats)."""

    def __init__(self, mean, std, inplace=False):
        self.mean, self.std, self.inplace = mean, std, inplace
        self.demean = [-m / s for m, s in zip(mean, std)]
        self.destd = [1 / s for s in std]

    def __call__(self, tensor):
        m = torch.as_tensor(self.demean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        s = torch.as_tensor(self.destd, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        return torch.clamp((tensor - m) / s, 0.0, 1.0)
