"""CPU: the C ABI.  libwu_kernels.so loads without a GPU, exports every entry point that include/wu_kernels.h
declares, and the ctypes binding (wu/_lib.py) covers exactly that set.  No compute is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wu_kernels.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from wu import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from wu import _build
        _build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in wu_kernels.h but not exported: {missing}"


def test_binding_matches_header():
    from wu import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.wu_version() >= 1
    assert lib.wu_last_error() is not None


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens on the host before any launch: a bad shape is rejected with a message."""
    from wu import _lib
    lib = _lib.load()
    rc = lib.wu_conv3x3_fwd(None, 0, None, None, None, 0, 1, 8, 8, 40, 64, 1, 0, None, 0, 0, None, 0, 0, _lib.BF16, None)
    assert rc < 0 and b"Cin" in lib.wu_last_error()
    assert lib.wu_conv3x3_wgrad_workspace(2, 16, 16, 48, 64, 1, _lib.BF16) == 0      # unsupported shape -> 0 bytes
    assert lib.wu_conv3x3_wgrad_workspace(2, 16, 32, 64, 64, 1, _lib.BF16) > 0


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the drop-in modules raise instead of computing on the host."""
    import torch
    import cunet
    import disc
    net = cunet.Conditional_UNet(5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 16, 16), torch.zeros(1, 5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        disc.SNDisc(5)(torch.zeros(1, 3, 16, 16), torch.zeros(1, 5))
    with pytest.raises(ValueError):
        cunet.Conditional_UNet(5, precision="fp16")
