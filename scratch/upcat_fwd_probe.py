"""Stand-alone launches of the fused AdaIN / bilinear x2 / dropout forward at the step's three levels (B=32, 256x256 bf16) with variants that take
pieces of its work away: p = 0 (no draws, no mask bytes), caller-supplied masks (no draws, mask bytes READ).  Timing by hipEvents, 4 back-to-back
launches; run under scratch/pmc_kernel.sh adain_upcat_fwd for the SQ counters."""
import os, statistics, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
import torch
from wu import kernels as K
from wu.layout import empty_nhwc
B, S = 32, 256
dev, bf = torch.device("cuda:0"), torch.bfloat16


def run(fn, reps=7, inner=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return statistics.median(ts)


for (c, h, cs) in [(128, S // 2, 64), (256, S // 4, 128), (512, S // 8, 256)]:
    x = (torch.rand((B, h, h, c), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
    cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, bf, dev)
    ys, ym = torch.rand((B, c), device=dev) + 0.5, torch.rand((B, c), device=dev)
    st = K.adain_stats(x, 1e-5)
    lo, hi = B * h * h * c * 2, B * 4 * h * h * c * 2
    t_drop = run(lambda: K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True))
    t_nomask = run(lambda: K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, False))
    t_p0 = run(lambda: K.adain_upcat(x, st, ys, ym, cat, 0.0, 123, False))
    mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
    t_maskin = run(lambda: K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, False, mask_in=mb))
    dense = empty_nhwc(B, c, 2 * h, 2 * h, bf, dev)
    t_dense = run(lambda: K.adain_upcat(x, st, ys, ym, dense, 0.3, 123, True))
    print(f"C={c} {h}->{2*h}: p=0.3 + mask bytes {t_drop:.1f} us ({(lo+hi+hi//16)/t_drop/1e6:.2f} TB/s) | p=0.3 no mask bytes {t_nomask:.1f} | p=0 {t_p0:.1f} us ({(lo+hi)/t_p0/1e6:.2f} TB/s) | "
          f"masks supplied (read, no draws) {t_maskin:.1f} | dense output (ld = C) {t_dense:.1f}", flush=True)
