"""SURVEY.md 8f.3 -- the GPU input pipeline (wu/input_pipeline.py, csrc/image.hip) against the Pillow chain the reference's
torchvision transforms run on the host (oracle/input_ref.py; t_cls_train.py:81-108), same random draws on both sides.
Bar: BIT-EXACT (integer / byte work): resize, rotation, flip, colour jitter and the float normalisation."""
import numpy as np
import pytest
import torch

from oracle import input_ref as IR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(sizes, seed):
    rng = np.random.default_rng(seed)
    hmax, wmax = max(h for h, _ in sizes), max(w for _, w in sizes)
    buf = np.zeros((len(sizes), hmax, wmax, 3), dtype=np.uint8)
    imgs = []
    for i, (h, w) in enumerate(sizes):
        # smooth structure + noise: exercises the resample filter and the enhancers' clipping
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(xx / 7.0 + i), 128 + 90 * np.cos(yy / 5.0), 40 + (xx + yy) % 200], -1)
        img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
        imgs.append(img)
        buf[i, :h, :w] = img
    return imgs, torch.from_numpy(buf).to(DEV)


SIZES = [(375, 500), (500, 333), (224, 224), (97, 131), (600, 800), (64, 48)]


@pytest.mark.parametrize("S", [224, 256, 64])
def test_train_transform_without_augmentation(S):
    """Resize((S, S)) -> RandomRotation(10) -> RandomHorizontalFlip -> ToTensor -> Normalize (t_cls_train.py:95-101)."""
    from wu.input_pipeline import GPUInputPipeline
    imgs, src = _batch(SIZES, 1)
    pipe = GPUInputPipeline(S, augmentation=False, seed=5)
    params = pipe.draw(SIZES)
    out = pipe(src, SIZES, params)
    assert tuple(out.shape) == (len(SIZES), 3, S, S) and out.dtype == torch.float32
    for i, (img, p) in enumerate(zip(imgs, params)):
        ref = IR.train_transform(img, S, p["angle"], p["flip"])
        assert np.array_equal(out[i].cpu().numpy(), ref), f"image {i} {img.shape}: max diff {np.abs(out[i].cpu().numpy() - ref).max()}"


@pytest.mark.parametrize("S", [224, 96])
def test_train_transform_with_augmentation(S):
    """RandomRotation -> RandomResizedCrop -> flip -> ColorJitter(0.5, 0.3, 0.3, 0) -> ToTensor -> Normalize (:81-93)."""
    from wu.input_pipeline import GPUInputPipeline
    imgs, src = _batch(SIZES, 2)
    pipe = GPUInputPipeline(S, augmentation=True, seed=11)
    for rep in range(2):
        params = pipe.draw(SIZES)
        out = pipe(src, SIZES, params)
        for i, (img, p) in enumerate(zip(imgs, params)):
            ref = IR.train_transform(img, S, p["angle"], p["flip"], True, p["crop"], p["factors"], p["order"])
            got = out[i].cpu().numpy()
            assert np.array_equal(got, ref), f"rep {rep} image {i}: {p}: max diff {np.abs(got - ref).max()}, {np.mean(got != ref):.4f} of pixels"


def test_test_transform_and_batch_of_one():
    from wu.input_pipeline import GPUInputPipeline
    imgs, src = _batch(SIZES, 3)
    out = GPUInputPipeline(128, train=False)(src, SIZES)
    for i, img in enumerate(imgs):
        assert np.array_equal(out[i].cpu().numpy(), IR.test_transform(img, 128))
    # upscaling (source smaller than S) and a single image
    imgs1, src1 = _batch([(40, 56)], 4)
    pipe = GPUInputPipeline(224, augmentation=False, seed=1)
    p = pipe.draw([(40, 56)])
    assert np.array_equal(pipe(src1, [(40, 56)], p)[0].cpu().numpy(), IR.train_transform(imgs1[0], 224, p[0]["angle"], p[0]["flip"]))
    with pytest.raises(ValueError):
        pipe(src1, [(41, 56)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pipe(src1.cpu(), [(40, 56)])


def test_feeds_the_generator():
    """End to end: uint8 batch -> GPU transforms -> Conditional_UNet forward (the (N,3,S,S) fp32 NCHW tensor in [-1, 1] the
    reference's loaders hand to the model, t_cls_train.py:414)."""
    import cunet
    from oracle import cunet_ref as O
    from wu.input_pipeline import GPUInputPipeline
    sizes = [(120, 160), (96, 96)]
    _, src = _batch(sizes, 6)
    x = GPUInputPipeline(64, augmentation=True, seed=2)(src, sizes)
    net = cunet.Conditional_UNet(5, precision="fp32")
    net.load_state_dict(O.make_cunet_params(5, 0))
    net = net.to(DEV).eval()
    c = torch.eye(5, device=DEV)[:2]
    with torch.no_grad():
        out = net(x, c)
    ref = O.cunet_forward(O.make_cunet_params(5, 0), x.cpu(), c.cpu())
    assert (out.cpu() - ref).abs().max().item() <= 1e-3
