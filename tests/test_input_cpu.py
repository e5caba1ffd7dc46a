"""CPU: the input pipeline's host logic and the pins of its oracle (oracle/input_ref.py) against Pillow itself -- the library the
reference's torchvision transforms call (t_cls_train.py:81-108).  The HIP kernels are compared with the same Pillow chain in
tests/test_gpu_input.py."""
import numpy as np
import pytest
from PIL import Image

from oracle import input_ref as IR


def _nearest_rotate_fixed(img, angle):
    """numpy restatement of libImaging/Geometry.c affine_fixed with the coefficients of oracle.input_ref.pil_rotate_coeffs."""
    h, w, _ = img.shape
    a = IR.pil_rotate_coeffs(angle, w, h)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xx = (a[2] + a[0] * xs + a[1] * ys)
    yy = (a[5] + a[3] * xs + a[4] * ys)
    xx = ((xx + 2 ** 31) % 2 ** 32) - 2 ** 31
    yy = ((yy + 2 ** 31) % 2 ** 32) - 2 ** 31
    xin, yin = xx >> 16, yy >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


@pytest.mark.parametrize("hw", [(37, 53), (64, 64), (101, 40)])
def test_rotate_coefficients_reproduce_pil(hw):
    """The product ships these coefficients to the GPU (wu.input_pipeline.rotate_coeffs is the same function): they must make
    Geometry.c's fixed-point nearest map reproduce Image.rotate bit for bit, including the black fill outside."""
    from wu.input_pipeline import rotate_coeffs
    rng = np.random.default_rng(hw[0])
    img = rng.integers(0, 256, (*hw, 3), dtype=np.uint8)
    for angle in (-10.0, -3.3, 0.25, 7.77, 10.0, 123.4):
        assert rotate_coeffs(angle, hw[1], hw[0]) == IR.pil_rotate_coeffs(angle, hw[1], hw[0])
        ref = np.asarray(Image.fromarray(img, "RGB").rotate(angle, Image.NEAREST, False, None))
        assert np.array_equal(_nearest_rotate_fixed(img, angle), ref), angle


def test_parameter_draws_follow_torchvision_get_params():
    from wu.input_pipeline import GPUInputPipeline
    sizes = [(375, 500), (600, 400), (32, 32)] * 40
    p = GPUInputPipeline(224, augmentation=True, seed=3)
    params = p.draw(sizes)
    angles = np.array([q["angle"] for q in params])
    assert angles.min() >= -10 and angles.max() <= 10 and angles.std() > 3           # RandomRotation(10): U(-10, 10)
    assert 0.3 < np.mean([q["flip"] for q in params]) < 0.7                          # RandomHorizontalFlip p = 0.5
    for (h, w), q in zip(sizes, params):
        i, j, ch, cw = q["crop"]                                                     # RandomResizedCrop: inside the image,
        assert 0 <= i and 0 <= j and i + ch <= h and j + cw <= w and ch >= 1 and cw >= 1
        assert 0.08 * 0.9 <= ch * cw / (h * w) <= 1.0 and 0.7 <= cw / ch <= 1.45 or (ch, cw) == (h, w)      # area / aspect ranges
        b, c, s = q["factors"]
        assert 0.5 <= b <= 1.5 and 0.7 <= c <= 1.3 and 0.7 <= s <= 1.3               # ColorJitter(0.5, 0.3, 0.3)
        assert sorted(q["order"]) == [0, 1, 2]
    assert len({q["order"] for q in params}) == 6                                    # shuffled per image
    plain = GPUInputPipeline(224, augmentation=False, seed=3).draw(sizes[:3])
    assert all(q["crop"] == (0, 0, h, w) and q["order"] == (-1, -1, -1) for (h, w), q in zip(sizes[:3], plain))
    test = GPUInputPipeline(224, train=False).draw(sizes[:3])
    assert all(q["angle"] == 0.0 and not q["flip"] for q in test)
    assert GPUInputPipeline(224, augmentation=True, seed=3).draw(sizes) == params    # seeded -> reproducible


def test_oracle_chain_shapes_and_range():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    out = IR.train_transform(img, 32, 4.0, True)
    assert out.shape == (3, 32, 32) and out.dtype == np.float32 and -1.0 <= out.min() and out.max() <= 1.0
    aug = IR.train_transform(img, 32, -6.0, False, True, (5, 8, 30, 40), (1.2, 0.8, 1.1), (2, 0, 1))
    assert aug.shape == (3, 32, 32)
    assert np.array_equal(IR.test_transform(img, 32), IR.to_tensor_normalize(np.asarray(Image.fromarray(img, "RGB").resize((32, 32), Image.BILINEAR))))
