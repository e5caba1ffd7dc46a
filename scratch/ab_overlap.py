"""Can an MFMA-bound weight-gradient kernel and an HBM-bound glue kernel really SHARE compute units (complementary resources), or do
they only take turns?  conv3x3_wgrad_v2 in its 8-wave shape (two waves per SIMD, ~454 of a SIMD's 512 registers, all 160 KiB of LDS)
and its 4-wave shape (one wave per SIMD, ~264 registers, same LDS) beside LDS-free glue kernels on a second stream: pair wall time
against the two stand-alone times, same box, interleaved."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc


def main():
    dev, bf, B = torch.device("cuda:0"), torch.bfloat16, 32
    def act(c, s):
        return (torch.rand((B, s, s, c), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
    # weight gradient of up2.0 (384 -> 128 @128): ~360 us
    x, gy = act(384, 128), act(128, 128)
    dw, db = torch.empty((128, 384, 3, 3), device=dev), torch.empty(128, device=dev)
    # glue: max-pool backward at level 1 (no LDS), AdaIN backward apply (no LDS), head backward
    xs, gs, gp = act(64, 256), act(64, 256), act(64, 128)
    dxx = empty_nhwc(B, 64, 256, 256, bf, dev)
    s2 = torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)

    def wg(shape):
        _lib.call("wu_set_option", 2, shape)
        K.conv3x3_wgrad(x, gy, dw, db)

    def pool():
        K.maxpool2_bwd(xs, gp, dxx, gs, 1)

    def timed(fa, fb, sa, sb, reps=3):
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(cur); sa.wait_event(e0); sb.wait_event(e0)
        for _ in range(reps):
            if fa is not None:
                with torch.cuda.stream(sa): fa()
            if fb is not None:
                with torch.cuda.stream(sb): fb()
        ea.record(sa); eb.record(sb); torch.cuda.synchronize()
        return max(e0.elapsed_time(ea), e0.elapsed_time(eb)) / reps * 1e3

    for _ in range(2):
        wg(1); wg(2); pool()
    torch.cuda.synchronize()
    res = {k: [] for k in ("w8", "w4", "pool", "w8+pool", "w4+pool", "w8;pool")}
    for _ in range(7):
        res["w8"].append(timed(lambda: wg(1), None, cur, s2))
        res["w4"].append(timed(lambda: wg(2), None, cur, s2))
        res["pool"].append(timed(pool, None, cur, s2))
        res["w8;pool"].append(timed(lambda: (wg(1), pool()), None, cur, s2))
        res["w8+pool"].append(timed(lambda: wg(1), pool, cur, s2))
        res["w4+pool"].append(timed(lambda: wg(2), pool, cur, s2))
    _lib.call("wu_set_option", 2, 1)
    for k, v in res.items():
        print(f"{k:10s} {statistics.median(v):8.1f} us")


if __name__ == "__main__":
    main()
