"""Stand-alone timing of the stem kernels at B = 32 / 64, 256 x 256 (4 back-to-back launches / 4, median of 7)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, resnet as RN  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402

dev = torch.device("cuda:0")
for B in (32, 64):
    S = 256
    x = torch.rand((B, 3, S, S), device=dev) * 2 - 1
    ws = (torch.rand((64, 3, 7, 7), device=dev) - 0.5) * 0.1
    bs = torch.rand(64, device=dev) - 0.5
    y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
    gy = (torch.rand((B, S // 2, S // 2, 64), device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    dx = torch.empty_like(x)

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
    for name, fn in (("stem fwd", lambda: RN.stem7x7(x, ws, bs, y, 1, _lib.BF16)), ("stem dgrad", lambda: RN.stem7x7_dgrad(gy, ws, dx, _lib.BF16))):
        fn()
        torch.cuda.synchronize()
        print(f"B={B} {name:12s} {timed(fn):8.1f} us")
