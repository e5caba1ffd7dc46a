"""Building blocks of the hot path -- same names and structure as the reference's nets.py.

``r_double_conv(cin, cout)``  (reference nets.py:18-24)  Conv3x3+ReLU, Conv3x3+ReLU
``sn_double_conv(cin, cout)`` (reference nets.py:26-33)  SN-Conv3x3 s1, SN-Conv3x3 s2, LeakyReLU(0.2)

Both return an indexable ``nn.Sequential`` whose children carry the reference's state-dict keys
(``0.weight``/``0.bias``/``2.weight``/``2.bias``; ``{0,1}.{bias,weight_orig,weight_u,weight_v}``), but
whose forward runs the fused HIP kernels of libwu_kernels.so (activation fused into the conv
epilogue; the ReLU / LeakyReLU children are structural placeholders).  Tensors flowing between blocks
are NHWC-strided (logical NCHW shape) in the compute dtype; an NCHW fp32 input is converted on entry.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F_

from wu import functional as WF
from wu import _lib
from wu.layout import precision_code, require_cuda

_DEFAULT_PRECISION = "bf16"


class _Conv3x3Base(nn.Module):
    """Shared forward of the plain and spectral-norm 3x3 convs."""

    def __init__(self, in_channels, out_channels, stride=1, act=WF.ACT_NONE):
        super().__init__()
        self.in_channels, self.out_channels, self.stride, self.act = in_channels, out_channels, stride, act
        self.kernel_size, self.padding = (3, 3), (1, 1)
        self._packed = WF.PackedConv()
        self.precision = _DEFAULT_PRECISION

    @staticmethod
    def _init_weight_bias(in_channels, out_channels):
        # nn.Conv2d's default init: the reference leaves init_weight() commented out (cunet.py:41)
        w = torch.empty(out_channels, in_channels, 3, 3)
        b = torch.empty(out_channels)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * 9)
        nn.init.uniform_(b, -bound, bound)
        return w, b

    def effective_weight(self):
        raise NotImplementedError

    def weight_ident(self):
        """Identity of what effective_weight() was computed from, for the packed-operand cache (None: the weight tensor's own
        storage pointer + version is the identity)."""
        return None

    def forward(self, x, act=None, out=None, out_nchw=False):
        require_cuda(x, type(self).__name__)
        act = self.act if act is None else act
        code = precision_code(self.precision)
        w = self.effective_weight()
        if self.in_channels == 3:
            return WF.conv3x3_c3(x, w, self.bias, self.stride, act, out_nchw, code)
        x = WF.to_nhwc(x, code)
        return WF.conv3x3(x, w, self.bias, self._packed, self.stride, act, out, self.weight_ident())

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size=(3, 3), stride={self.stride}, padding=(1, 1)"


class Conv3x3(_Conv3x3Base):
    """nn.Conv2d(cin, cout, 3, padding=1, stride=stride) with nn.Conv2d's parameters (OIHW fp32
    ``weight``, ``bias``) and a fused activation epilogue."""

    def __init__(self, in_channels, out_channels, stride=1, act=WF.ACT_NONE):
        super().__init__(in_channels, out_channels, stride, act)
        w, b = self._init_weight_bias(in_channels, out_channels)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)

    def effective_weight(self):
        return self.weight


class SNConv3x3(_Conv3x3Base):
    """nn.utils.spectral_norm(nn.Conv2d(...)) (reference nets.py:28-31): parameters ``bias`` +
    ``weight_orig``, buffers ``weight_u`` / ``weight_v``; one power iteration per training forward (buffers
    updated in place), W = weight_orig / sigma with sigma = u . (W_mat v)."""

    def __init__(self, in_channels, out_channels, stride=1, act=WF.ACT_NONE, eps=1e-12):
        super().__init__(in_channels, out_channels, stride, act)
        w, b = self._init_weight_bias(in_channels, out_channels)
        self.eps = eps
        self.bias = nn.Parameter(b)
        self.weight_orig = nn.Parameter(w)
        with torch.no_grad():
            u = F_.normalize(w.new_empty(out_channels).normal_(0, 1), dim=0, eps=eps)
            v = F_.normalize(w.new_empty(in_channels * 9).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)
        self._sn_generation = 0        # bumped whenever a power iteration rewrites weight_u / weight_v through raw pointers
        self._w_eff_next = None        # W/sigma handed in by the owner's batched normalisation (disc.SNDisc.forward), used once

    @property
    def weight(self):
        """The normalised weight for the current buffers (readable, as disc.py:16-19 requires)."""
        return spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, False, self.eps)

    def effective_weight(self):
        w = self._w_eff_next
        if w is not None:              # normalised by the owner together with its other SN layers (one batched call)
            self._w_eff_next = None
            return w
        if self.training:
            self._sn_generation += 1
        return spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, self.training, self.eps)

    def weight_ident(self):
        # W/sigma is a fresh temporary per forward (its pointer repeats from step to step): key the packed operands on what it
        # was computed from -- weight_orig (optimizer steps bump its version), the power-iteration buffers (load_state_dict
        # bumps theirs; the HIP power iteration writes them through raw pointers, hence the explicit generation)
        return (self.weight_orig.data_ptr(), self.weight_orig._version, self.weight_u.data_ptr(), self.weight_u._version,
                self.weight_v._version, self._sn_generation)


def spectral_normalize(weight_orig, u, v, do_power_iteration, eps=1e-12):
    """torch.nn.utils.spectral_norm.compute_weight.  On the GPU: the fused HIP kernels (5 launches instead of ~25 tiny
    torch kernels); on CPU tensors (module construction / state-dict inspection only): the same arithmetic in torch."""
    if weight_orig.is_cuda:
        return WF.spectral_normalize(weight_orig, u, v, do_power_iteration, eps)
    w_mat = weight_orig.reshape(weight_orig.shape[0], -1)
    if do_power_iteration:
        with torch.no_grad():
            v_new = F_.normalize(torch.mv(w_mat.t(), u), dim=0, eps=eps)
            u_new = F_.normalize(torch.mv(w_mat, v_new), dim=0, eps=eps)
            v.copy_(v_new)
            u.copy_(u_new)
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return weight_orig / sigma


class SNLinear(nn.Module):
    """nn.utils.spectral_norm(nn.Linear(cin, cout)) (reference disc.py:21-24, the heads ``l`` and ``embed``) with the state-dict
    keys torch gives it -- ``bias``, ``weight_orig``, ``weight_u``, ``weight_v`` -- and the same RNG draws at construction
    (nn.Linear's init, then u ~ N(0,1)^cout and v ~ N(0,1)^cin, normalised), on the HIP spectral-norm kernels: torch's hook is ~25
    launches per forward and layer, this shares the network's one batched normalisation (SNDisc.forward)."""

    def __init__(self, in_features, out_features, bias=True, eps=1e-12):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=bias)
        self.in_features, self.out_features, self.eps = in_features, out_features, eps
        self.bias = lin.bias
        self.weight_orig = nn.Parameter(lin.weight.detach())
        with torch.no_grad():
            u = F_.normalize(self.weight_orig.new_empty(out_features).normal_(0, 1), dim=0, eps=eps)
            v = F_.normalize(self.weight_orig.new_empty(in_features).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)
        self._sn_generation = 0
        self._w_eff_next = None

    @property
    def weight(self):
        return spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, False, self.eps)

    def effective_weight(self):
        w = self._w_eff_next
        if w is not None:
            self._w_eff_next = None
            return w
        if self.training:
            self._sn_generation += 1
        return spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, self.training, self.eps)

    def forward(self, x):
        return F_.linear(x, self.effective_weight(), self.bias)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class _FusedSequential(nn.Sequential):
    def set_precision(self, precision):
        for m in self.modules():
            if isinstance(m, _Conv3x3Base):
                m.precision = precision
        return self


class RDoubleConv(_FusedSequential):
    """reference nets.py:18-24."""

    def forward(self, x, out=None):
        x = self[0](x, act=WF.ACT_RELU)
        return self[2](x, act=WF.ACT_RELU, out=out)


class SNDoubleConv(_FusedSequential):
    """reference nets.py:26-33: no activation between the two convs, LeakyReLU(0.2) after the second."""

    def forward(self, x):
        nchw_mid = self[0].in_channels == 3       # 3->3 conv keeps the image layout (NCHW fp32)
        x = self[0](x, act=WF.ACT_NONE, out_nchw=nchw_mid)
        return self[1](x, act=WF.ACT_LEAKY)


def r_double_conv(in_channels, out_channels):
    return RDoubleConv(
        Conv3x3(in_channels, out_channels),
        nn.ReLU(inplace=True),
        Conv3x3(out_channels, out_channels),
        nn.ReLU(inplace=True),
    )


def sn_double_conv(in_channels, out_channels):
    return SNDoubleConv(
        SNConv3x3(in_channels, in_channels),
        SNConv3x3(in_channels, out_channels, stride=2),
        nn.LeakyReLU(0.2, inplace=True),
    )


def upsample_box(out_channels):
    """reference nets.py:4-8 -- dead code there (never called); kept importable, stock torch modules."""
    return nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                         nn.BatchNorm2d(out_channels, affine=False))


def double_conv(in_channels, out_channels):
    """reference nets.py:10-16 -- dead code there (never called); kept importable, stock torch modules."""
    return nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1),
                         nn.Conv2d(in_channels, out_channels, 3, padding=1, stride=2),
                         nn.BatchNorm2d(out_channels, affine=False), nn.LeakyReLU(0.2, inplace=True))
