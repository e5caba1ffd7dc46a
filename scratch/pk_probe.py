"""Is the cross-stream hazard of DESIGN.md 4 a property of packed-FP32 FMAs as such?  A register-only kernel (scratch/micro/pk_probe.hip:
v_pk_fma_f32 or v_fmac_f32 in a loop, no LDS, no memory traffic inside) on a second stream BESIDE one other kernel on the main stream;
its output compared bitwise with the same launch alone.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/micro/pk_probe.hip -o scratch/micro/libpk_probe.so   (build container)
    python scratch/pk_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K, resnet as RN  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402
from wu.unet_graph import _side_stream  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "scratch", "micro", "libpk_probe.so"))
lib.pk_probe.restype = ctypes.c_int
lib.pk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
CODE = _lib.BF16
B, S = 64, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
ws = ((torch.rand((64, 3, 7, 7), generator=g) - 0.5) * 0.1).to(dev)
bs = (torch.rand(64, generator=g) - 0.5).to(dev)
stem_y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
gy_stem = (torch.rand((B, S // 2, S // 2, 64), generator=g) - 0.5).to(torch.bfloat16).to(dev).permute(0, 3, 1, 2)
dx_stem = torch.empty_like(x)
pool_y = empty_nhwc(B, 256, S // 4, S // 4, torch.bfloat16, dev)
w11 = ((torch.rand((1024, 256), generator=g) - 0.5) * 0.1).to(torch.bfloat16).to(dev)
pw_y = empty_nhwc(B, 1024, S // 4, S // 4, torch.bfloat16, dev)
xc = (torch.rand((B, S // 4, S // 4, 256), generator=g) - 0.5).to(torch.bfloat16).to(dev).permute(0, 3, 1, 2)
wc = (torch.rand((256, 256, 3, 3), generator=g) - 0.5).to(dev) * 0.05
wcf, _ = K.pack_conv3x3(wc, CODE)
yc = empty_nhwc(B, 256, S // 4, S // 4, torch.bfloat16, dev)
NF = 256 * 16 * 4096                      # 16 M floats: 4096 workgroups
src = (torch.rand(NF, generator=g) * 2 - 1).to(dev)
main, side = torch.cuda.current_stream(dev), _side_stream(dev)
ITERS = 2000                              # ~100 us of FMAs per launch

others = {
    "nothing": lambda: None,
    "stem7x7 fwd (MFMA, VGPR accumulators)": lambda: RN.stem7x7(x, ws, bs, stem_y, 1, CODE),
    "stem7x7 dgrad (MFMA)": lambda: RN.stem7x7_dgrad(gy_stem, ws, dx_stem, CODE),
    "conv1x1 256->1024 (MFMA)": lambda: RN.conv1x1(pool_y, w11, None, pw_y, 1),
    "conv3x3 256->256 @64 (persistent MFMA, full LDS)": lambda: K.conv3x3(xc, wcf, None, yc, 1, 1),
    "torch elementwise": lambda: torch.add(x, 1.0),
}
for packed in (1, 0):
    ref = torch.empty_like(src)
    assert lib.pk_probe(ref.data_ptr(), src.data_ptr(), NF, ITERS, packed, main.cuda_stream) == 0
    torch.cuda.synchronize()
    for oname, other in others.items():
        bad, nwrong = 0, 0
        for rep in range(8):
            out = torch.full_like(src, float("nan"))
            torch.cuda.synchronize()
            torch.cuda._sleep(3_000_000)
            side.wait_stream(main)
            lib.pk_probe(out.data_ptr(), src.data_ptr(), NF, ITERS, packed, side.cuda_stream)
            for _ in range(2):
                other()
            main.wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                nwrong = max(nwrong, int(((out != ref) | out.isnan()).sum().item()))
        print(f"{'v_pk_fma_f32' if packed else 'v_fmac_f32 '} loop beside {oname:50s}: {bad}/8 launches differ" + (f" (up to {nwrong} elements)" if bad else ""))
