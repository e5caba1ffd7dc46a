"""adain_upcat_bwd at the three decoder levels (B=32, bf16, keep bits, ReLU-gated dx): round-3 marching kernel (option 8 = 5) against the
round-4 LDS-ring kernel + row-bound apply (option 8 = 1), interleaved, with per-kernel times from HIP events around each variant."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc

dev, bf, B = torch.device("cuda:0"), torch.bfloat16, int(os.environ.get("B", "32"))
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
tot = {5: 0.0, 1: 0.0}
for (c, h, cs) in [(128, 128, 64), (256, 64, 128), (512, 32, 256)]:
    x = act(c, h); cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, bf, dev); gc = act(c + cs, 2 * h)
    ys = torch.rand((B, c), device=dev) + 0.5; ym = torch.rand((B, c), device=dev)
    st = K.adain_stats(x, 1e-5); dx = empty_nhwc(B, c, h, h, bf, dev)
    mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
    t = {}
    for rnd in range(2):
        for mode in (5, 1):
            _lib.call("wu_set_option", 8, mode)
            t.setdefault(mode, []).append(run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1)))
    _lib.call("wu_set_option", 8, 1)
    mbytes = (B * 4 * h * h * c * 2 * (1 + 1 / 16) + 5 * B * h * h * c * 2) / 1e6        # dy + keep bytes once; x twice, g' out and in, dx out
    a, b = min(t[5]), min(t[1])
    tot[5] += a; tot[1] += b
    print(f"C={c} {2*h}->{h}: marching {a:7.1f} us ({mbytes / a * 1e3:.0f} GB/s)   LDS ring {b:7.1f} us ({mbytes / b * 1e3:.0f} GB/s, {100 * mbytes / b * 1e3 / 8000:.0f} % of 8 TB/s)   algorithmic {mbytes:.0f} MB", flush=True)
print(f"sum: marching {tot[5]:.1f} us, LDS ring {tot[1]:.1f} us")
