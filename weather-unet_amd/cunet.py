"""Conditional U-Net generator -- drop-in for the reference's cunet.py (cunet.py:7-82).

Same constructor, same ``forward(x, c) -> (N,3,H,W)`` fp32 tanh image, same 39 state-dict keys in the
same OIHW shapes, ``.parameters()`` usable by torch.optim.Adam, ``.train()/.eval()`` toggling the
Dropout(p=0.3) -- but every op on the path is a hand-written HIP kernel for gfx950:

  encoder / decoder r_double_conv  -> fused conv3x3+bias+ReLU MFMA implicit GEMM (first conv: thin direct)
  maxpool                          -> NHWC 16-B/lane kernel
  adain -> upsample -> dropout -> cat  -> ONE fused kernel writing channels [0,C) of a concat buffer whose
                                         channels [C,..) were written in place by the encoder conv (no cat copy)
  conv_last + tanh                 -> fused 1x1 kernel writing the NCHW fp32 image

``precision``: "bf16" (default; bf16 storage + bf16 MFMA, fp32 accumulate; <= 5e-2 max-abs of the
reference) or "fp32" (fp32 storage, exact-fp32 MFMA; <= 1e-3).
"""
import itertools

import torch
import torch.nn as nn

from nets import r_double_conv
from utils import AdaIN, BatchNorm, HalfDropout  # noqa: F401  (names the reference imports, cunet.py:3)
from wu import functional as WF
from wu.layout import empty_nhwc, precision_code, require_cuda, torch_dtype
from wu.unet_graph import unet_forward

_SEED_COUNTER = itertools.count(1)


class Conditional_UNet(nn.Module):

    def init_weight(self, std=0.2):
        """reference cunet.py:9-16 (defined, never called there: :41 is commented out)."""
        with torch.no_grad():      # in place on the parameters themselves (not .data): bumps ._version -> packed operands follow
            for m in self.modules():
                cn = m.__class__.__name__
                if cn.find('Conv') != -1 and hasattr(m, 'weight') and isinstance(m.weight, nn.Parameter):
                    m.weight.normal_(0., std)
                elif cn.find('Linear') != -1:
                    m.weight.normal_(1., std)
                    m.bias.fill_(0)

    def __init__(self, num_classes, precision="bf16"):
        super(Conditional_UNet, self).__init__()

        self.dconv_down1 = r_double_conv(3, 64)
        self.dconv_down2 = r_double_conv(64, 128)
        self.dconv_down3 = r_double_conv(128, 256)
        self.dconv_down4 = r_double_conv(256, 512)

        # structural twins of cunet.py:26-28 (parameter-free; their work happens in the fused kernels)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.maxpool = nn.MaxPool2d(2)
        self.dropout = nn.Dropout(p=0.3)

        self.adain3 = AdaIN(512, num_classes=num_classes)
        self.adain2 = AdaIN(256, num_classes=num_classes)
        self.adain1 = AdaIN(128, num_classes=num_classes)

        self.dconv_up3 = r_double_conv(256 + 512, 256)
        self.dconv_up2 = r_double_conv(128 + 256, 128)
        self.dconv_up1 = r_double_conv(64 + 128, 64)

        self.conv_last = nn.Conv2d(64, 3, 1)
        self.activation = nn.Tanh()
        self.set_precision(precision)
        self.dropout_seed = None     # int -> reproducible dropout masks (tests); None -> fresh seed per call
        self.dropout_masks = None    # (m3, m2, m1) NCHW keep-masks for cunet.py:61,68,75: used INSTEAD of the counter RNG
        self._seed_dev = None        # int64 device scalar added to the seeds inside the kernels (wu.graph_infer: per-replay masks)
        self.fused = True            # one autograd node for the whole net (wu/unet_graph.py); False = per-layer Functions
        self.grad_sink = None        # wu.ddp.GradBucketReducer.attach(): overlap gradient all-reduce with the fused backward

    def set_precision(self, precision):
        precision_code(precision)
        self.precision = precision
        for m in self.children():
            if hasattr(m, "set_precision"):
                m.set_precision(precision)
            elif isinstance(m, AdaIN):
                m.precision = precision
        return self

    def _next_seed(self, k):
        if self.dropout_seed is not None:
            return (int(self.dropout_seed) * 4 + k) & 0x7FFFFFFFFFFFFFFF
        base = torch.initial_seed() & 0xFFFFFFFF
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        return ((base << 24) ^ (rank << 56) ^ next(_SEED_COUNTER)) & 0x7FFFFFFFFFFFFFFF

    def _up(self, k, adain, x, c, skip, catbuf):
        """cunet.py:59-62 / 66-69 / 73-76 as one fused op."""
        y_std, y_mean = adain.style(c)
        p = self.dropout.p if self.training else 0.0
        mask_in = None
        if p > 0 and self.dropout_masks is not None:
            from wu.kernels import pack_keep_mask
            mask_in = pack_keep_mask(self.dropout_masks[3 - k].to(x.device), x.dtype)
        return WF.adain_upcat(x, y_std, y_mean, skip, catbuf, adain.eps, p, self._next_seed(k), self._seed_dev, mask_in)

    def forward(self, x, c, encoder_cache=None):
        """``encoder_cache`` (fused graph only; an extension, not in the reference's signature): a dict the caller passes to TWO
        forwards of the same ``x`` between which the weights do not change -- the encoder (cunet.py:45-54: no Dropout, independent of
        ``c``) is then computed by the first and reused by the second, bit for bit what two full forwards give."""
        require_cuda(x, "Conditional_UNet")
        n, _, h, w = x.shape
        if h % 8 or w % 8:
            raise ValueError(f"Conditional_UNet: H and W must be divisible by 8 (three 2x poolings), got {h}x{w}")
        if self.fused:
            return unet_forward(self, x, c, encoder_cache)
        code = precision_code(self.precision)
        dt, dev = torch_dtype(code), x.device
        c = c.to(device=dev, dtype=torch.float32)

        # concat buffers: [upsampled | skip] (torch.cat order of cunet.py:62,69,76); the encoder convs write
        # their outputs straight into the skip slices
        cat1 = empty_nhwc(n, 128 + 64, h, w, dt, dev)
        cat2 = empty_nhwc(n, 256 + 128, h // 2, w // 2, dt, dev)
        cat3 = empty_nhwc(n, 512 + 256, h // 4, w // 4, dt, dev)

        conv1 = self.dconv_down1(x, out=cat1[:, 128:])       # :45
        x = WF.maxpool2(conv1)                               # :46
        conv2 = self.dconv_down2(x, out=cat2[:, 256:])       # :48
        x = WF.maxpool2(conv2)                               # :49
        conv3 = self.dconv_down3(x, out=cat3[:, 512:])       # :51
        x = WF.maxpool2(conv3)                               # :52
        x = self.dconv_down4(x)                              # :54

        x = self._up(3, self.adain3, x, c, conv3, cat3)      # :59-62
        x = self.dconv_up3(x)                                # :64
        x = self._up(2, self.adain2, x, c, conv2, cat2)      # :66-69
        x = self.dconv_up2(x)                                # :71
        x = self._up(1, self.adain1, x, c, conv1, cat1)      # :73-76
        x = self.dconv_up1(x)                                # :78

        return WF.conv1x1_tanh(x, self.conv_last.weight, self.conv_last.bias)   # :80-82
