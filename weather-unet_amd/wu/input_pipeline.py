"""GPU-side input pipeline: the transforms the reference applies per image on the host before every step
(t_cls_train.py:81-108; t_est_train.py:76-103 is the same block), on a whole batch of decoded uint8 images resident in HBM.

    train, no --augmentation :  Resize((S, S)) -> RandomRotation(10) -> RandomHorizontalFlip -> ToTensor -> Normalize(.5, .5)
    train, --augmentation    :  RandomRotation(10) -> RandomResizedCrop(S) -> RandomHorizontalFlip -> ColorJitter(.5, .3, .3, 0)
                                -> ToTensor -> Normalize
    test                     :  Resize((S, S)) -> ToTensor -> Normalize

JPEG decoding stays on the host (dataset.py:63,91: PIL ``Image.open``); what arrives here is the decoded RGB batch, one padded
``(N, Hmax, Wmax, 3)`` uint8 tensor plus the true ``(h, w)`` of every image.  The kernels (csrc/image.hip) reproduce Pillow's
arithmetic bit for bit -- two-pass fixed-point bilinear resample, 16.16 fixed-point nearest rotation, ImageEnhance blends -- so
the output equals what the reference's DataLoader workers would have produced for the same random draws; the draws themselves
(``draw``) follow torchvision 0.3's ``get_params`` (uniform angle, log-uniform aspect ratio with 10 attempts and the central-crop
fallback, jitter factors and a shuffled op order) from a seeded ``random.Random``.
"""
import math
import random

import numpy as np
import torch

from . import _lib
from .layout import require_cuda, stream_ptr

BRIGHTNESS, CONTRAST, SATURATION = 0, 1, 2


def rotate_coeffs(angle_deg, w, h):
    """PIL Image.rotate(angle, NEAREST, expand=False, center=None) as the six 16.16 fixed-point coefficients of
    libImaging/Geometry.c affine_fixed (matrix about (w/2, h/2), FIX(v) = floor(v * 65536 + 0.5), pixel centres at +0.5)."""
    a = -math.radians(angle_deg % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)
    out = [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]
    return [((v + 2 ** 31) % 2 ** 32) - 2 ** 31 for v in out]       # C int


class GPUInputPipeline:
    def __init__(self, input_size, augmentation=False, train=True, seed=None, degrees=10.0,
                 brightness=0.5, contrast=0.3, saturation=0.3, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.S = int(input_size)
        self.augmentation, self.train = bool(augmentation), bool(train)
        self.degrees, self.scale, self.ratio = float(degrees), scale, ratio
        self.jitter = (brightness, contrast, saturation)
        self.rng = random.Random(seed)
        self._ws = None

    # ---- the random draws of torchvision 0.3's transforms (host side, a few numbers per image) ----
    def draw(self, sizes):
        """One parameter set per image: dict(angle, flip, crop (i, j, h, w), factors (b, c, s), order)."""
        out = []
        for (h, w) in sizes:
            p = {"angle": 0.0, "flip": False, "crop": (0, 0, int(h), int(w)), "factors": (1.0, 1.0, 1.0), "order": (-1, -1, -1)}
            if self.train:
                if self.augmentation:
                    p["angle"] = self.rng.uniform(-self.degrees, self.degrees)             # RandomRotation.get_params
                    p["crop"] = self._resized_crop_params(int(h), int(w))                  # RandomResizedCrop.get_params
                    p["flip"] = self.rng.random() < 0.5                                     # RandomHorizontalFlip
                    b, c, s = self.jitter                                                   # ColorJitter.get_params
                    p["factors"] = (self.rng.uniform(max(0.0, 1 - b), 1 + b), self.rng.uniform(max(0.0, 1 - c), 1 + c),
                                    self.rng.uniform(max(0.0, 1 - s), 1 + s))
                    order = [BRIGHTNESS, CONTRAST, SATURATION]
                    self.rng.shuffle(order)
                    p["order"] = tuple(order)
                else:
                    p["angle"] = self.rng.uniform(-self.degrees, self.degrees)
                    p["flip"] = self.rng.random() < 0.5
            out.append(p)
        return out

    def _resized_crop_params(self, height, width):
        area = height * width
        for _ in range(10):
            target_area = self.rng.uniform(*self.scale) * area
            aspect = math.exp(self.rng.uniform(math.log(self.ratio[0]), math.log(self.ratio[1])))
            w = int(round(math.sqrt(target_area * aspect)))
            h = int(round(math.sqrt(target_area / aspect)))
            if 0 < w <= width and 0 < h <= height:
                return (self.rng.randint(0, height - h), self.rng.randint(0, width - w), h, w)
        in_ratio = width / height                                                           # fallback: central crop
        if in_ratio < min(self.ratio):
            w = width
            h = int(round(w / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            h = height
            w = int(round(h * max(self.ratio)))
        else:
            w, h = width, height
        return ((height - h) // 2, (width - w) // 2, h, w)

    # ---- the batch transform ----
    def __call__(self, src_u8, sizes, params=None):
        """src_u8: (N, Hmax, Wmax, 3) uint8 CUDA tensor; sizes: N x (h, w); returns (N, 3, S, S) float32 in [-1, 1]."""
        require_cuda(src_u8, "GPUInputPipeline")
        if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3 or not src_u8.is_contiguous():
            raise ValueError("GPUInputPipeline: expected a contiguous (N, Hmax, Wmax, 3) uint8 tensor")
        n, hmax, wmax, _ = src_u8.shape
        if len(sizes) != n or any(h < 1 or w < 1 or h > hmax or w > wmax for h, w in sizes):
            raise ValueError("GPUInputPipeline: sizes must give one (h, w) <= (Hmax, Wmax) per image")
        if params is None:
            params = self.draw(sizes)
        S, dev = self.S, src_u8.device
        rot_first = self.train and self.augmentation
        geo = np.zeros((n, 18), dtype=np.int32)
        fac = np.ones((n, 3), dtype=np.float32)
        order = np.full((n, 3), -1, dtype=np.int32)
        ksize, jitter = 3, False
        for i, ((h, w), p) in enumerate(zip(sizes, params)):
            ct, cl, ch, cw = p["crop"]
            if ct < 0 or cl < 0 or ch < 1 or cw < 1 or ct + ch > h or cl + cw > w:
                raise ValueError(f"GPUInputPipeline: crop {p['crop']} outside image {(h, w)}")
            geo[i, 0:2] = np.array([i * hmax * wmax * 3], dtype=np.int64).view(np.int32)
            geo[i, 2:9] = (h, w, wmax, ct, cl, ch, cw)
            geo[i, 9] = 1 if p["flip"] else 0
            do_rot = self.train and p["angle"] % 360.0 != 0.0
            if do_rot:
                geo[i, 10:16] = rotate_coeffs(p["angle"], w if rot_first else S, h if rot_first else S)
            geo[i, 16] = 1 if do_rot else 0
            ksize = max(ksize, 2 * int(math.ceil(max(ch / S, cw / S, 1.0))) + 1)
            fac[i] = p["factors"]
            order[i] = p["order"]
            jitter = jitter or any(o >= 0 for o in p["order"])
        lib = _lib.load()
        assert lib.wu_image_geo_bytes() == 72
        geo_d = torch.from_numpy(geo).to(dev)
        nbytes = lib.wu_image_workspace_bytes(n, S, ksize)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        if not jitter:
            _lib.call("wu_image_geometry", src_u8.data_ptr(), geo_d.data_ptr(), self._ws.data_ptr(), self._ws.numel(), None, out.data_ptr(),
                      n, S, ksize, 1 if rot_first else 0, stream_ptr())
            return out
        stage = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev)
        _lib.call("wu_image_geometry", src_u8.data_ptr(), geo_d.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stage.data_ptr(), None,
                  n, S, ksize, 1 if rot_first else 0, stream_ptr())
        fac_d, ord_d = torch.from_numpy(fac).to(dev), torch.from_numpy(order).to(dev)
        _lib.call("wu_image_color_jitter", stage.data_ptr(), fac_d.data_ptr(), ord_d.data_ptr(), out.data_ptr(), n, S, stream_ptr())
        return out
