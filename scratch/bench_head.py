"""Last decoder conv + 1x1 head + tanh: two launches against the fused launch (csrc/conv3x3_mfma_v2.hip GATED = 5), with and without the
64-channel output.  B = 32 256 x 256 (the training step) and B = 16 512 x 512 (BASELINE configs[4]); 4 back-to-back launches / 4, median of 7,
two passes (the first timings of a process read slow: profiles/r04_down12_probe.txt), ReLU-distributed input like the step feeds it."""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402

dev = torch.device("cuda:0")


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


for B, S in ((32, 256), (16, 512)):
    torch.manual_seed(0)
    x = torch.relu(torch.randn((B, S, S, 64), device=dev)).to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = torch.randn((64, 64, 3, 3), device=dev) * (2.0 / 576) ** 0.5
    bias = torch.rand(64, device=dev) * 0.1
    hw_ = torch.randn((3, 64), device=dev) * 0.2
    hb = torch.rand(3, device=dev) * 0.1
    wf, _ = K.pack_conv3x3(wt, _lib.BF16)
    y = empty_nhwc(B, 64, S, S, torch.bfloat16, dev)
    out = torch.empty((B, 3, S, S), device=dev)
    cases = [
        ("conv + ReLU", lambda: K.conv3x3(x, wf, bias, y, 1, K.ACT_RELU)),
        ("head kernel", lambda: K.conv1x1_tanh(y, hw_, hb, out)),
        ("conv + ReLU, then head", lambda: (K.conv3x3(x, wf, bias, y, 1, K.ACT_RELU), K.conv1x1_tanh(y, hw_, hb, out))),
        ("fused, y written", lambda: K.conv3x3_relu_head(x, wf, bias, y, hw_, hb, out)),
        ("fused, no y", lambda: K.conv3x3_relu_head(x, wf, bias, None, hw_, hb, out)),
    ]
    for p in range(2):
        for name, fn in cases:
            fn()
            torch.cuda.synchronize()
            t = timed(fn)
            if p == 1:
                print(f"B={B} {S}x{S}  {name:26s} {t:8.1f} us", flush=True)
