"""Spectral-norm projection discriminator -- drop-in for the reference's disc.py (disc.py:8-38).

4 x (SN-conv3x3 s1 -> SN-conv3x3 s2 -> LeakyReLU 0.2) -> global SUM pool -> SN-Linear(512,1) +
<SN-Linear(nc,512)(c), feat>.  ``forward(x, c) -> [out(N,1), c1, c2, c3, c4]``; 40 state-dict keys
(10 SN layers x bias / weight_orig / weight_u / weight_v).  The convs reuse the generator's HIP kernels
(stride-1/2 MFMA implicit GEMM, LeakyReLU epilogue); the two linear heads are (N,512) GEMVs (F.linear) on spectrally normalised weights of the same HIP kernels (nets.SNLinear).  All ten spectral
normalisations of a forward (one power iteration each in training mode, as torch's hooks do) run as ONE batched call before the convs.
The feature maps c1..c4 are returned as NHWC-strided tensors of the compute dtype (logical NCHW shape).
"""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from nets import SNLinear, sn_double_conv
from utils import ConditionalNorm  # noqa: F401  (imported by the reference, disc.py:5)
from wu import disc_graph as DG
from wu import functional as WF
from wu import kernels as K
from wu.layout import precision_code, require_cuda


# one batched spectral normalisation per forward (default) or one per layer (A/B switch, WU_SN_BATCHED=0)
BATCHED_SPECTRAL_NORM = os.environ.get("WU_SN_BATCHED", "1") == "1"


# the conv trunk as one autograd node (default) or one node per layer (A/B switch, WU_DISC_FUSED=0)
FUSED_TRUNK = os.environ.get("WU_DISC_FUSED", "1") == "1"


_SN_DONE = weakref.WeakKeyDictionary()


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

        self.l = SNLinear(512, 1)                                            # nn.utils.spectral_norm(nn.Linear(512, 1)), disc.py:21
        nn.init.xavier_uniform_(self.l.weight_orig)

        self.embed = SNLinear(num_classes, 512, bias=True)                   # disc.py:23
        nn.init.xavier_uniform_(self.embed.weight_orig)
        self.set_precision(precision)

    def set_precision(self, precision):
        precision_code(precision)
        self.precision = precision
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            m.set_precision(precision)
        return self

    @property
    def sn_done(self):
        """Event recorded (on the stream of the forward) once the last forward's batched normalisation was enqueued; None before
        the first one.  Kept outside the module's attributes: events cannot be deep-copied or pickled."""
        return _SN_DONE.get(self)

    def sn_layers(self):
        """The ten spectrally normalised layers in forward order."""
        return [m for i in range(1, 5) for m in getattr(self, 'conv{}'.format(i))[:2]] + [self.l, self.embed]

    def _normalize_weights(self):
        """W/sigma of all ten SN layers in one batched call (what torch's ten forward pre-hooks compute one by one: a power
        iteration per layer in training mode, buffers updated in place), handed to the layers for their next forward; the packed
        MFMA operands of the six wide convs are rebuilt in one launch as well."""
        layers = self.sn_layers()
        self.sn_batched_last = False                         # did THIS forward normalise in one batched call (and record sn_done)?
        if not BATCHED_SPECTRAL_NORM or len({m.training for m in layers}) != 1 or len({m.eps for m in layers}) != 1 or not layers[0].weight_orig.is_cuda:
            return []                                        # mixed modes / CPU: every layer normalises its own weight
        train = layers[0].training
        w_eff = WF.spectral_normalize_multi([m.weight_orig for m in layers], [m.weight_u for m in layers], [m.weight_v for m in layers],
                                            train, layers[0].eps)
        for m, w in zip(layers, w_eff):
            if train:
                m._sn_generation += 1
            m._w_eff_next = w
        code = precision_code(self.precision)
        todo = [(m, w) for m, w in zip(layers[:8], w_eff[:8]) if m.in_channels != 3 and m._packed.stale(w, code, m.weight_ident())]
        if todo:
            for (m, w), (wf, wd) in zip(todo, K.pack_conv3x3_multi([w for _, w in todo], code)):
                m._packed.w_fwd, m._packed.w_dgrad = wf, wd
                m._packed.key = m._packed.make_key(w, code, m.weight_ident())
        # Recorded AFTER the pack launch (round 4, advisor): in eval mode the cache key does not change between the two passes of
        # update_discriminator, so a pass on the other stream finds the operands "fresh" and waits for this event only -- it must
        # cover the kernel that writes them, not just the power iteration (wu/train_step.py).
        ev = _SN_DONE.get(self)
        if ev is None:
            ev = _SN_DONE[self] = torch.cuda.Event()
        ev.record()                                          # buffers AND packed operands are final for this forward
        self.sn_batched_last = True
        return layers

    def forward(self, x, c=None):
        require_cuda(x, "SNDisc")
        pending = self._normalize_weights()
        try:
            return self._forward(x, c)
        finally:
            for m in pending:                                # an exception mid-way must not leave a weight behind for a later call
                m._w_eff_next = None

    def _forward(self, x, c):
        if FUSED_TRUNK and x.is_cuda:
            # :28-32 as ONE autograd node (wu/disc_graph.py): same kernels, one static schedule, LeakyReLU gates in the data-gradient
            # epilogues; bit-identical to the per-layer path below
            convs = [m for i in range(1, 5) for m in getattr(self, 'conv{}'.format(i))[:2]]
            ws = [m.effective_weight() for m in convs]       # handed in by the batched normalisation, or normalised here layer by layer
            x, c1, c2, c3, c4 = DG.trunk(x, ws, [m.bias for m in convs], [(m._packed, m.weight_ident()) for m in convs[2:]], self.precision)
        else:
            c1 = self.conv1(x)                               # :28
            c2 = self.conv2(c1)                              # :29
            c3 = self.conv3(c2)                              # :30
            c4 = self.conv4(c3)                              # :31
            x = WF.sumpool(c4)                               # :32 global pool (sum over H, W), fp32 (N,512)
        out = self.l(x)                                      # :33
        e_c = self.embed(c.to(device=x.device, dtype=torch.float32))   # :34 (c=None raises here, as in the reference)
        if c is not None:
            out = out + torch.sum(e_c * x, dim=1, keepdim=True)        # :36
        return [out, c1, c2, c3, c4]
