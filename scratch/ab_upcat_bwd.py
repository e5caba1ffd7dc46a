"""adain_upcat_bwd at the three decoder levels (B=32): marching with two columns per thread (option 8 = 3) with stored keep-bits (p = 0.3)
and without dropout (p = 0), the 16-tap gather formulation (8 = 0), marching with one column per thread (8 = 2)."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc

dev, bf, B = torch.device("cuda:0"), torch.bfloat16, 32
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
for (c, h, cs) in [(128, 128, 64), (256, 64, 128), (512, 32, 256)]:
    x = act(c, h); cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, bf, dev); gc = act(c + cs, 2 * h)
    ys = torch.rand((B, c), device=dev) + 0.5; ym = torch.rand((B, c), device=dev)
    st = K.adain_stats(x, 1e-5); dx = empty_nhwc(B, c, h, h, bf, dev)
    mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
    _lib.call("wu_set_option", 8, 3)
    t_mask = run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1))
    ref = dx.clone()
    _lib.call("wu_set_option", 8, 2)
    t_one = run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1))
    rel = ((dx.float() - ref.float()).norm() / ref.float().norm()).item()
    _lib.call("wu_set_option", 8, 1)
    t_nodrop = run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.0, 123, None, 1))
    _lib.call("wu_set_option", 8, 0)
    t_gather = run(lambda: K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1))
    _lib.call("wu_set_option", 8, 1)
    mbytes = (B * 4 * h * h * c * 2 * (1 + 1 / 16) + 2 * B * h * h * c * 2) / 1e6
    print(f"C={c} {2*h}->{h}: march+bits {t_mask:7.1f} us ({mbytes / t_mask * 1e3:.0f} GB/s)   march, no dropout {t_nodrop:7.1f} us   gather+bits {t_gather:7.1f} us   march 1 col/thread {t_one:7.1f} us (rel diff {rel:.1e})")
