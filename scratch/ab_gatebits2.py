"""A/B: forward conv with / without gate-bit output for the shapes that produce bits in the cUNet (same box).
    python scratch/ab_gatebits2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch
from wu import kernels as K
from wu.layout import empty_nhwc, precision_code


def bench(fn, reps=9):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


dev = torch.device("cuda")
code = precision_code("bf16")
n = 32
for (ci, co, h, kind) in [(64, 64, 256, "randn"), (64, 64, 256, "rand"), (64, 128, 128, "rand"), (128, 256, 64, "rand"), (256, 512, 32, "rand"),
                          (768, 256, 64, "rand"), (384, 128, 128, "rand"), (192, 64, 256, "rand")]:
    torch.manual_seed(0)
    x = empty_nhwc(n, ci, h, h, torch.bfloat16, dev)
    x.copy_(torch.randn(n, ci, h, h, device=dev) if kind == "randn" else torch.rand(n, ci, h, h, device=dev) * 2 - 1)
    w = (torch.rand(co, ci, 3, 3, device=dev) * 2 - 1) * 0.05
    wf, wd = K.pack_conv3x3(w, code)
    out = empty_nhwc(n, co, h, h, torch.bfloat16, dev)
    bits = K.gate_bits_alloc(out)
    b = torch.zeros(co, device=dev)
    for _ in range(2):
        K.conv3x3(x, wf, b, out, 1, 1); K.conv3x3_bits(x, wf, b, out, 1, gate_bits_out=bits)
    t0 = bench(lambda: K.conv3x3(x, wf, b, out, 1, 1))
    t1 = bench(lambda: K.conv3x3_bits(x, wf, b, out, 1, gate_bits_out=bits))
    t2 = bench(lambda: K.conv3x3(x, wf, b, out, 1, 1))
    print(f"{ci}->{co} @{h} ({kind}): fwd {t0:.1f} / {t2:.1f} us, fwd+bits {t1:.1f} us")
