"""r03_layer_table.md reads 202 us for down1.2 forward (64 -> 64 @256, + fused pool) where the same round's PMC pass reads 154 us
and the stamps 162 us.  Which harness property does it: the kernel variant (fused pool), the input distribution (the table feeds U(-1,1); in
the step the input is a ReLU output, half zeros), the output stride (the step writes the concat slice, ld 192), or the position (first kernel
timed after start-up)?  Every combination, the first one measured twice (cold, then again at the end)."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import kernels as K
from wu.layout import empty_nhwc
dev, bf, B, S = torch.device("cuda:0"), torch.bfloat16, 32, 256
def run(fn, reps=7, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return statistics.median(ts)
w = (torch.rand((64, 64, 3, 3), device=dev) * 2 - 1) * 0.05
wf, _ = K.pack_conv3x3(w, 1)
bias = torch.zeros(64, device=dev)
xu = (torch.rand((B, S, S, 64), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
xr = torch.relu(torch.randn((B, S, S, 64), device=dev)).to(bf).permute(0, 3, 1, 2)
y_dense = empty_nhwc(B, 64, S, S, bf, dev)
cat = empty_nhwc(B, 192, S, S, bf, dev)
y_slice = cat[:, 128:]
pool = empty_nhwc(B, 64, S // 2, S // 2, bf, dev)
cases = []
for xn, x in (("U(-1,1)", xu), ("ReLU(N(0,1))", xr)):
    for yn, y in (("dense y", y_dense), ("concat slice", y_slice)):
        for kn, fn in (("conv+pool", lambda x=x, y=y: K.conv3x3_relu_pool(x, wf, bias, y, pool)), ("plain conv", lambda x=x, y=y: K.conv3x3(x, wf, bias, y, 1, 1))):
            cases.append((f"{xn:13s} {yn:13s} {kn}", fn))
print(f"COLD (first timing of the process): {cases[0][0]}: {run(cases[0][1]):.1f} us")
for name, fn in cases:
    print(f"{name}: {run(fn):.1f} us", flush=True)
print(f"again, the first case: {run(cases[0][1]):.1f} us")
