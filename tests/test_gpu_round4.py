"""Round-4 GPU tests: kernels and schedules added this round, each against the kernel / order it replaces or against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


# ---------------------------------------------------------------------------------------------------------------------------------
# AdaIN / bilinear x2 / dropout backward through the LDS ring (csrc/glue.hip: adain_upcat_bwd_tile_kernel, utils.py:41-51 + cunet.py:59-62)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [
    (2, 128, 24, 40, 64),      # two column tiles (the second one 8 columns wide), three strips, ragged right edge inside a tile
    (1, 64, 7, 33, 64),        # odd height, 33 columns: a one-column second tile; single strip
    (3, 256, 16, 32, 128),     # exactly one full column tile: the right halo pixel is outside the image
    (2, 64, 40, 70, 0),        # three column tiles, five strips, no skip channels behind the slice
    (1, 512, 2, 2, 256),       # smallest legal image
])
@pytest.mark.parametrize("p_drop", [0.3, 0.0])
def test_adain_upcat_bwd_lds_ring_vs_marching(shape, p_drop):
    """The LDS-ring kernel computes g' with the marching kernel's arithmetic term for term: the parked g' must be BIT-IDENTICAL; the
    per-(n, c) sums are grouped by other tiles (fp32 re-association) and dx follows from them within bf16 rounding."""
    from wu import _lib, kernels as K
    from wu.layout import as_nhwc, empty_nhwc, nhwc_ld, stream_ptr
    n, c, h, w, cs = shape
    dev = _dev()
    bf = torch.bfloat16
    x = as_nhwc(_rand((n, c, h, w), 81, -1, 2).to(dev), _lib.BF16)
    g = as_nhwc(_rand((n, c + cs, 2 * h, 2 * w), 82).to(dev), _lib.BF16)
    ystd = _rand((n, c), 83, 0.5, 1.5).to(dev)
    ymean = _rand((n, c), 84).to(dev)
    stats = K.adain_stats(x, 1e-5)
    cat = empty_nhwc(n, c + cs, 2 * h, 2 * w, bf, dev)
    mb = K.adain_upcat(x, stats, ystd, ymean, cat, p_drop, 4711, True)
    res = {}
    try:
        for mode in (5, 1):
            _lib.call("wu_set_option", 8, mode)
            dx = torch.full((n, h, w, c), float("nan"), dtype=bf, device=dev).permute(0, 3, 1, 2)
            dstd, dmean = torch.empty((n, c), device=dev), torch.empty((n, c), device=dev)
            gtmp = torch.full((n, h, w, c), float("nan"), dtype=bf, device=dev)
            sums = torch.empty((n, c, 2 * (1 + K.MAX_SPLITS)), device=dev)
            _lib.call("wu_adain_upcat_bwd", g.data_ptr(), nhwc_ld(g), x.data_ptr(), nhwc_ld(x), stats.data_ptr(), ystd.data_ptr(),
                      dx.data_ptr(), nhwc_ld(dx), dstd.data_ptr(), dmean.data_ptr(), gtmp.data_ptr(), sums.data_ptr(),
                      n, h, w, c, float(p_drop), 4711, mb.data_ptr() if mb is not None else None, 1, _lib.BF16, stream_ptr())
            torch.cuda.synchronize()
            res[mode] = (gtmp.float().clone(), dstd.clone(), dmean.clone(), dx.float().clone())
    finally:
        _lib.call("wu_set_option", 8, 1)
    old, new = res[5], res[1]
    assert not torch.isnan(new[0]).any() and not torch.isnan(new[3]).any()
    assert torch.equal(old[0], new[0]), f"parked g' differs: {(old[0] - new[0]).abs().max().item()}"
    for a, b, name in ((old[1], new[1], "d y_std"), (old[2], new[2], "d y_mean")):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()) * (h * w) ** 0.5, name
    assert (old[3] - new[3]).abs().max().item() <= 1.6e-2 * max(1.0, old[3].abs().max().item())
