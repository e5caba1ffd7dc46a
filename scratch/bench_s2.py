"""SNDisc's stride-2 convs (nets.py:29-30) at B=32 / 64, bf16: forward (LeakyReLU), data gradient, weight gradient -- us and TFLOP/s per
launch (4 back-to-back launches per timing, median of 7)."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc

dev, bf = torch.device("cuda:0"), torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def act(c, s):
    return (torch.rand((B, s, s, c), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
def run(fn, reps=7, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return statistics.median(ts)
tot = [0.0, 0.0, 0.0]
for cin, s in ((64, 128), (128, 64), (256, 32)):
    cout, so = 2 * cin, s // 2
    x, gy = act(cin, s), act(cout, so)
    w = (torch.rand((cout, cin, 3, 3), device=dev) - 0.5) * 0.1
    bias = torch.rand(cout, device=dev)
    wf, wd = K.pack_conv3x3(w, _lib.BF16)
    y, dx = empty_nhwc(B, cout, so, so, bf, dev), empty_nhwc(B, cin, s, s, bf, dev)
    dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
    gf = 2.0 * B * so * so * 9 * cin * cout / 1e9
    t = [run(lambda: K.conv3x3(x, wf, bias, y, 2, K.ACT_LEAKY)), run(lambda: K.conv3x3_s2_dgrad(gy, wd, dx)), run(lambda: K.conv3x3_wgrad(x, gy, dw, db, 2))]
    for i in range(3): tot[i] += t[i]
    print(f"{cin:3d} -> {cout:3d} @{s:3d}->{so:3d} B={B} ({gf:5.1f} GFLOP): fwd {t[0]:6.1f} us {gf / t[0] * 1e3:5.0f} TFLOP/s | dgrad {t[1]:6.1f} us {gf / t[1] * 1e3:5.0f} | wgrad {t[2]:6.1f} us {gf / t[2] * 1e3:5.0f}", flush=True)
print(f"sum: fwd {tot[0]:.1f} us, dgrad {tot[1]:.1f} us, wgrad {tot[2]:.1f} us")
