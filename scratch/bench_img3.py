"""Stand-alone timings of the discriminator's image-layout 3 -> 3 conv (disc.conv1[0], NCHW fp32 in and out) and the 3 -> 64 stride-2
conv behind it: forward, weight gradient, data gradient at B = 32, 256 x 256 (4 back-to-back launches / 4, median of 7)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402

dev = torch.device("cuda:0")
B, S = 32, 256
x = torch.rand((B, 3, S, S), device=dev) * 2 - 1
w33 = (torch.rand((3, 3, 3, 3), device=dev) - 0.5) * 0.3
b3 = torch.rand(3, device=dev) - 0.5
y33 = torch.empty((B, 3, S, S), device=dev)
g33 = torch.rand((B, 3, S, S), device=dev) - 0.5
dw33, db3, dx = torch.empty_like(w33), torch.empty(3, device=dev), torch.empty_like(x)
w64 = (torch.rand((64, 3, 3, 3), device=dev) - 0.5) * 0.3
b64 = torch.rand(64, device=dev) - 0.5
y64 = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
g64 = (torch.rand((B, S // 2, S // 2, 64), device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
dw64, db64 = torch.empty_like(w64), torch.empty(64, device=dev)


def timed(fn):
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    return statistics.median(ts)


cases = [
    ("3->3 s1 fwd (NCHW fp32)", lambda: K.conv3x3_c3(x, w33, b3, y33, 1, 0, True, _lib.BF16)),
    ("3->3 s1 wgrad", lambda: K.conv3x3_c3_wgrad(x, g33, dw33, db3, 1, _lib.BF16, dy_nchw=True)),
    ("3->3 s1 dgrad", lambda: K.conv3x3_c3_dgrad(g33, w33, dx, 1, _lib.BF16, dy_nchw=True)),
    ("3->64 s2 fwd leaky", lambda: K.conv3x3_c3(y33, w64, b64, y64, 2, 2, False, _lib.BF16)),
    ("3->64 s2 wgrad (gated)", lambda: K.conv3x3_c3_wgrad(y33, g64, dw64, db64, 2, _lib.BF16, y=y64, act=2)),
    ("3->64 s2 dgrad (gated)", lambda: K.conv3x3_c3_dgrad(g64, w64, dx, 2, _lib.BF16, y=y64, act=2)),
]
for name, fn in cases:
    fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {timed(fn):8.1f} us")
