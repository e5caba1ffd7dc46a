"""Minimal reproducer attempt for the cross-stream hazard: the register-only / LDS-fed instruction shapes of
scratch/micro/pk_probe2.hip on a second stream BESIDE the stem's MFMA kernel, each compared bitwise with the same launch alone.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/micro/pk_probe2.hip -o scratch/micro/libpk_probe2.so   (build container)
    python scratch/pk_probe2.py"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, resnet as RN  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402
from wu.unet_graph import _side_stream  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "scratch", "micro", "libpk_probe2.so"))
lib.pk_probe2.restype = ctypes.c_int
lib.pk_probe2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
CODE = _lib.BF16
B, S = 64, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
ws = ((torch.rand((64, 3, 7, 7), generator=g) - 0.5) * 0.1).to(dev)
bs = (torch.rand(64, generator=g) - 0.5).to(dev)
stem_y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
NF = 256 * 4 * 8192
src = (torch.rand(NF, generator=g) * 2 - 1).to(dev)
main, side = torch.cuda.current_stream(dev), _side_stream(dev)
ITERS = int(os.environ.get("ITERS", "1500"))
REPS = int(os.environ.get("REPS", "8"))
NAMES = {0: "v_mov lo halves -> v_pk_fma_f32 op_sel_hi:[1,0,1] (the compiled shape)", 1: "the same + s_nop 3 before the packed FMA",
         2: "all four halves written 16+ wait states ahead", 3: "shape 0 without the op_sel broadcast",
         4: "shape 0 as two v_fmac_f32 (control)", 5: "LDS-fed whole pair (ds_read_b64), lgkmcnt(0)", 6: "LDS-fed low halves (ds_read_b32), lgkmcnt(0)"}


def stem_twice():
    RN.stem7x7(x, ws, bs, stem_y, 1, CODE)
    RN.stem7x7(x, ws, bs, stem_y, 1, CODE)


others = {"nothing": (lambda: None, 2), "stem7x7_fwd_mfma x2": (stem_twice, REPS)}
print(f"{NF} floats, {ITERS} iterations per lane; launches differing from the same launch alone")
for variant, vname in NAMES.items():
    ref = torch.empty_like(src)
    assert lib.pk_probe2(ref.data_ptr(), src.data_ptr(), NF, ITERS, variant, main.cuda_stream) == 0
    torch.cuda.synchronize()
    parts = []
    for oname, (other, reps) in others.items():
        bad, nwrong, cols = 0, 0, None
        for rep in range(reps):
            out = torch.full_like(src, float("nan"))
            torch.cuda.synchronize()
            torch.cuda._sleep(3_000_000)
            side.wait_stream(main)
            assert lib.pk_probe2(out.data_ptr(), src.data_ptr(), NF, ITERS, variant, side.cuda_stream) == 0
            other()
            main.wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                wrong = ((out != ref) | out.isnan()).view(-1, 4)
                nwrong = max(nwrong, int(wrong.any(1).sum().item()))
                cols = wrong.sum(0).tolist()
        parts.append(f"beside {oname}: {bad}/{reps}" + (f" (<= {nwrong} lanes; wrong [acc.lo, acc.hi, side, t+u] = {cols})" if bad else ""))
    print(f"variant {variant} ({vname:72s}): " + " | ".join(parts), flush=True)
