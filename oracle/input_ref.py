"""CPU oracle for the input pipeline: the per-image transforms of t_cls_train.py:81-108, run with PIL itself.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference composes ``torchvision.transforms`` (Pipfile:11 pins torchvision<0.4), which are thin wrappers over Pillow:

    transforms.Resize((S, S))                  -> img.resize((S, S), Image.BILINEAR)
    transforms.RandomRotation(10)              -> img.rotate(angle, Image.NEAREST, expand=False, center=None), angle ~ U(-10, 10)
    transforms.RandomResizedCrop(S)            -> img.crop((j, i, j + w, i + h)).resize((S, S), Image.BILINEAR)
    transforms.RandomHorizontalFlip()          -> img.transpose(Image.FLIP_LEFT_RIGHT) with probability 0.5
    transforms.ColorJitter(0.5, 0.3, 0.3, 0)   -> ImageEnhance.Brightness / Contrast / Color (img).enhance(factor), random order
    transforms.ToTensor(), Normalize(.5, .5)   -> (uint8 / 255 - 0.5) / 0.5, CHW float32

torchvision is not importable here (no network), Pillow is (12.x): the oracle applies exactly those Pillow calls with
EXPLICIT parameters (angle, flip, crop box, factors, order), so the random draws are inputs, not part of what is compared.
Parity status: the Pillow calls are the real third-party code the reference runs; the torchvision wrapper layer (which call
it makes with which arguments, the parameter distributions of ``get_params``) is restated from torchvision 0.3's published
source and is UNPINNED against torchvision itself.

``pil_rotate_coeffs`` restates how Image.rotate / ImagingTransformAffine turn an angle into the 16.16 fixed-point coefficients of
Geometry.c:affine_fixed; it is checked against ``img.rotate`` itself in tests/test_input_cpu.py.
"""
import math

import numpy as np
from PIL import Image, ImageEnhance

BRIGHTNESS, CONTRAST, SATURATION = 0, 1, 2


def pil_rotate_coeffs(angle_deg, w, h):
    """Image.rotate(angle, expand=False, center=None) -> the six 16.16 fixed-point ints of Geometry.c affine_fixed
    (PIL/Image.py rotate(): matrix about the centre (w/2, h/2); libImaging/Geometry.c: FIX(v) = floor(v * 65536 + 0.5))."""
    angle = angle_deg % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def train_transform(img_u8, size, angle, flip, augmentation=False, crop=None, factors=None, order=None):
    """t_cls_train.py:81-101 on one HxWx3 uint8 array with the random draws given: returns (3, S, S) float32.
    augmentation=False: Resize -> RandomRotation -> RandomHorizontalFlip (:96-99);
    augmentation=True:  RandomRotation -> RandomResizedCrop(crop = (i, j, h, w)) -> flip -> ColorJitter (:82-91)."""
    img = Image.fromarray(img_u8, "RGB")
    if not augmentation:
        img = img.resize((size, size), Image.BILINEAR)
        img = img.rotate(angle, Image.NEAREST, False, None)
    else:
        img = img.rotate(angle, Image.NEAREST, False, None)
        i, j, h, w = crop
        img = img.crop((j, i, j + w, i + h)).resize((size, size), Image.BILINEAR)
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    if augmentation and order is not None:
        for op in order:
            if op == BRIGHTNESS:
                img = ImageEnhance.Brightness(img).enhance(factors[0])
            elif op == CONTRAST:
                img = ImageEnhance.Contrast(img).enhance(factors[1])
            elif op == SATURATION:
                img = ImageEnhance.Color(img).enhance(factors[2])
    return to_tensor_normalize(np.asarray(img))


def test_transform(img_u8, size):
    """t_cls_train.py:103-107: Resize -> ToTensor -> Normalize."""
    return to_tensor_normalize(np.asarray(Image.fromarray(img_u8, "RGB").resize((size, size), Image.BILINEAR)))


def to_tensor_normalize(hwc_u8):
    """transforms.ToTensor + Normalize(mean 0.5, std 0.5): float32 arithmetic as torch does it (div by 255, sub, div)."""
    t = hwc_u8.astype(np.float32) / np.float32(255.0)
    t = (t - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(t.transpose(2, 0, 1))
