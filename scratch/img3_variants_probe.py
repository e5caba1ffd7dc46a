"""The discriminating run for the cross-stream hazard: every form of scratch/micro/img3_variants.hip
(two builds x {plain, fmac1} x {compiler's queue, shallow, all reads up front}) on a second stream BESIDE the stem's MFMA kernel,
compared bitwise with the same launch alone.  One pass, REPS launches per form; output -> profiles/r04_hazard.txt.
    scratch/micro/build_img3_variants.sh      (build container)
    python scratch/img3_variants_probe.py     (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, resnet as RN  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402
from wu.unet_graph import _side_stream  # noqa: E402

REPS = int(os.environ.get("REPS", "10"))
dev = torch.device("cuda:0")
CODE = _lib.BF16
B, S = 32, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
w33 = ((torch.rand((3, 3, 3, 3), generator=g) - 0.5) * 0.3).to(dev)
b3 = (torch.rand(3, generator=g) - 0.5).to(dev)
ws = ((torch.rand((64, 3, 7, 7), generator=g) - 0.5) * 0.1).to(dev)
bs = (torch.rand(64, generator=g) - 0.5).to(dev)
stem_y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
RN.stem7x7(x, ws, bs, stem_y, 1, CODE)
torch.cuda.synchronize()
stem_ref = stem_y.clone()
main, side = torch.cuda.current_stream(dev), _side_stream(dev)
want = torch.nn.functional.conv2d(x.double().cpu(), w33.double().cpu(), b3.double().cpu(), padding=1).float().to(dev)

FORMS = {0: "plain FMA, compiler's LDS queue", 1: "plain FMA, <= 3 LDS reads in flight", 2: "plain FMA, all LDS reads up front",
         3: "fmac1, compiler's LDS queue", 4: "fmac1, <= 3 LDS reads in flight", 5: "fmac1, all LDS reads up front"}
print(f"B={B} {S}x{S}, {REPS} launches per form beside stem7x7_fwd_mfma_kernel (and 3 beside nothing)")
for build in ("packed", "scalar"):
    lib = ctypes.CDLL(os.path.join(ROOT, "scratch", "micro", f"libimg3v_{build}.so"))
    lib.img3v_conv.restype = ctypes.c_int
    lib.img3v_conv.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    for form, fname in FORMS.items():
        def subj(out, stream):
            rc = lib.img3v_conv(form, x.data_ptr(), w33.data_ptr(), b3.data_ptr(), out.data_ptr(), B, S, S, stream.cuda_stream)
            assert rc == 0, rc
        ref = torch.empty_like(x)
        subj(ref, main)
        torch.cuda.synchronize()
        alone_err = (ref - want).abs().max().item()
        line = []
        for oname, other, reps in (("nothing", lambda: None, 3), ("stem", lambda: RN.stem7x7(x, ws, bs, stem_y, 1, CODE), REPS)):
            bad, groups, worst = 0, 0, 0.0
            for rep in range(reps):
                out = torch.full_like(x, float("nan"))
                torch.cuda.synchronize()
                torch.cuda._sleep(3_000_000)               # ~1.5 ms blocker: both streams start together behind it
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    subj(out, side)
                other()
                main.wait_stream(side)
                torch.cuda.synchronize()
                assert torch.equal(stem_y, stem_ref), "the stem's own output changed"
                if not torch.equal(out, ref):
                    bad += 1
                    wrong = (out != ref) | out.isnan()
                    groups = max(groups, int(wrong.sum().item()))
                    worst = max(worst, (out - ref).abs().nan_to_num(1e9).max().item())
                    if bad == 1:
                        idx = wrong.nonzero()[:4].tolist()
                        first = "; first wrong (n, plane, h, w): " + str(idx)
            line.append(f"beside {oname}: {bad}/{reps} differ" + (f" (<= {groups} elements, max |diff| {worst:.3e}{first})" if bad else ""))
        print(f"{build:6s} build, form {form} ({fname:36s}) alone vs fp64 {alone_err:.1e} | " + " | ".join(line), flush=True)
