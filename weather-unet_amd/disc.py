"""Spectral-norm projection discriminator -- drop-in for the reference's disc.py (disc.py:8-38).

4 x (SN-conv3x3 s1 -> SN-conv3x3 s2 -> LeakyReLU 0.2) -> global SUM pool -> SN-Linear(512,1) +
<SN-Linear(nc,512)(c), feat>.  ``forward(x, c) -> [out(N,1), c1, c2, c3, c4]``; 40 state-dict keys
(10 SN layers x bias / weight_orig / weight_u / weight_v).  The convs reuse the generator's HIP kernels
(stride-1/2 MFMA implicit GEMM, LeakyReLU epilogue); the two linear heads are (N,512) GEMVs kept in torch.
The feature maps c1..c4 are returned as NHWC-strided tensors of the compute dtype (logical NCHW shape).
"""
import numpy as np
import torch
import torch.nn as nn

from nets import sn_double_conv
from utils import ConditionalNorm  # noqa: F401  (imported by the reference, disc.py:5)
from wu import functional as WF
from wu.layout import precision_code, require_cuda


class SNDisc(nn.Module):

    def __init__(self, num_classes, precision="bf16"):
        super().__init__()
        self.conv1 = sn_double_conv(3, 64)
        self.conv2 = sn_double_conv(64, 128)
        self.conv3 = sn_double_conv(128, 256)
        self.conv4 = sn_double_conv(256, 512)
        # reference disc.py:16-19 applies xavier_uniform_ to `.weight`.  At construction time that attribute is what
        # torch.nn.utils.spectral_norm registered: `weight_orig.data` -- a plain tensor SHARING weight_orig's storage -- so
        # the in-place init lands in weight_orig (measured against the reference: after `SNDisc(5)` every weight_orig lies
        # within the xavier bound, not the Conv2d default's; SURVEY 8a11's "no-op" reading does not hold on this torch).
        # Same RNG draws in the same order here: torch.manual_seed(s); SNDisc(nc) gives the reference's 40 state-dict
        # tensors bit for bit (tests/golden/sndisc_default_init_*.npz, captured from the reference itself).
        for i in range(1, 5):
            for j in range(2):
                nn.init.xavier_uniform_(getattr(self, 'conv{}'.format(i))[j].weight_orig, np.sqrt(2))

        self.l = nn.utils.spectral_norm(nn.Linear(512, 1))
        nn.init.xavier_uniform_(self.l.weight_orig)

        self.embed = nn.utils.spectral_norm(nn.Linear(num_classes, 512, bias=True))
        nn.init.xavier_uniform_(self.embed.weight_orig)
        self.set_precision(precision)

    def set_precision(self, precision):
        precision_code(precision)
        self.precision = precision
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            m.set_precision(precision)
        return self

    def forward(self, x, c=None):
        require_cuda(x, "SNDisc")
        c1 = self.conv1(x)                                   # :28
        c2 = self.conv2(c1)                                  # :29
        c3 = self.conv3(c2)                                  # :30
        c4 = self.conv4(c3)                                  # :31
        x = WF.sumpool(c4)                                   # :32 global pool (sum over H, W), fp32 (N,512)
        out = self.l(x)                                      # :33
        e_c = self.embed(c.to(device=x.device, dtype=torch.float32))   # :34 (c=None raises here, as in the reference)
        if c is not None:
            out = out + torch.sum(e_c * x, dim=1, keepdim=True)        # :36
        return [out, c1, c2, c3, c4]
