import os, sys, statistics
sys.path.insert(0, "weather-unet_amd")
import torch
from wu import resnet as RN
from wu.layout import empty_nhwc
dev = torch.device("cuda:0")
for B in (32, 64):
    x = torch.rand((B, 3, 256, 256), device=dev) * 2 - 1
    w = (torch.rand((64, 3, 7, 7), device=dev) - 0.5) * 0.2
    b = torch.rand(64, device=dev) - 0.5
    for code, dt in ((1, torch.bfloat16), (0, torch.float32)):
        y = empty_nhwc(B, 64, 128, 128, dt, dev)
        for _ in range(3): RN.stem7x7(x, w, b, y, 1, code)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): RN.stem7x7(x, w, b, y, 1, code)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        print(f"stem B={B} {'bf16' if code else 'fp32'}: {statistics.median(ts):.1f} us")
