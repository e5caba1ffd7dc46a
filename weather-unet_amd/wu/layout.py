"""NHWC tensor helpers.

Activations are torch tensors with the reference's LOGICAL shape (N, C, H, W) (cunet.py:43-82 never
sees anything else) but channels-last MEMORY: stride (H*W*ld, 1, W*ld, ld) where ld >= C is the pixel
stride of the buffer the tensor lives in (ld > C for a channel slice of a concat buffer).
"""
import torch

from . import _lib

_TORCH_DTYPE = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16}
_CODE = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def torch_dtype(code):
    return _TORCH_DTYPE[code]


def dtype_code(t):
    try:
        return _CODE[t.dtype]
    except KeyError:
        raise TypeError(f"hot-path tensors must be float32 or bfloat16, got {t.dtype}") from None


def precision_code(precision):
    if precision in ("bf16", "bfloat16", torch.bfloat16):
        return _lib.BF16
    if precision in ("fp32", "float32", torch.float32):
        return _lib.F32
    raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")


def empty_nhwc(n, c, h, w, dtype, device):
    """(N,C,H,W)-shaped tensor over a dense NHWC buffer."""
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def zeros_nhwc(n, c, h, w, dtype, device):
    return torch.zeros((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def nhwc_ld(t):
    """Pixel stride of an NHWC-strided tensor; raises if `t` is not NHWC-strided."""
    if t.dim() != 4:
        raise ValueError(f"expected a 4-D activation, got shape {tuple(t.shape)}")
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ld = sw
    ok = (sc == 1 or c == 1) and ld >= c and (sh == w * ld or h == 1) and (sn == h * w * ld or n == 1)
    if not ok:
        raise ValueError(f"tensor of shape {tuple(t.shape)} / stride {t.stride()} is not NHWC-strided")
    return ld


def is_nhwc(t):
    try:
        nhwc_ld(t)
        return t.is_cuda
    except ValueError:
        return False


def require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: the MI355X hot path runs HIP kernels only -- got a {t.device} tensor. "
            "There is no CPU fallback (the CPU oracle lives in oracle/ and is test-only).")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def as_nhwc(x, code):
    """Bring an arbitrary (N,C,H,W) CUDA tensor into NHWC storage of dtype `code`.  NCHW-contiguous fp32
    input goes through the transposing HIP kernel; an already-NHWC tensor of the right dtype is returned
    unchanged."""
    require_cuda(x, "as_nhwc")
    tdt = torch_dtype(code)
    if x.dtype == tdt and is_nhwc(x):
        return x
    n, c, h, w = x.shape
    if x.dtype == torch.float32 and x.is_contiguous():
        y = empty_nhwc(n, c, h, w, tdt, x.device)
        _lib.call("wu_nchw_f32_to_nhwc", x.data_ptr(), y.data_ptr(), c, n, h, w, c, code, stream_ptr())
        return y
    return x.to(tdt).contiguous(memory_format=torch.channels_last)


def to_nchw_f32(x):
    """NHWC hot-path tensor -> NCHW-contiguous fp32 (the reference's layout)."""
    require_cuda(x, "to_nchw_f32")
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    _lib.call("wu_nhwc_to_nchw_f32", x.data_ptr(), nhwc_ld(x), y.data_ptr(), n, h, w, c, dtype_code(x), stream_ptr())
    return y
