"""Is the estimator's pointwise GEMM bound by latency or by L2 -> CU bytes?  Time per launch against the number of GEMM rows, for the three
row-tile sizes of conv1x1_mfma_kernel (option 12): the slope (us per 8192 rows) against the L2 -> CU bytes the tiling implies.
    python scratch/pw_scaling.py"""
import os
import statistics
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import resnet as RN
    from wu import _lib
    from wu.layout import empty_nhwc
    dev = torch.device("cuda:0")
    bf = torch.bfloat16

    def run(fn, reps=9, inner=8):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / inner)
        return statistics.median(ts)

    for (ci, co, res) in [(256, 1024, True), (1024, 256, False), (512, 2048, True), (2048, 512, False), (128, 512, True), (512, 128, False)]:
        print(f"--- {ci} -> {co}{' +res' if res else ''}")
        for opt, tm in [(0, 64), (2, 128), (1, 256)]:
            _lib.call("wu_set_option", 12, opt)
            row = []
            for B in (8, 16, 32, 64, 128, 256):
                s = 16
                x = (torch.rand((B, s, s, ci), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2)
                w = ((torch.rand((co, ci), device=dev) - 0.5) * 0.1).to(bf)
                b = torch.zeros(co, device=dev)
                y = empty_nhwc(B, co, s, s, bf, dev)
                r = (torch.rand((B, s, s, co), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2) if res else None
                row.append(run(lambda: RN.conv1x1(x, w, b, y, 1, residual=r)))
            M = [B * 256 for B in (8, 16, 32, 64, 128, 256)]
            slope = (row[-1] - row[-2]) / ((M[-1] - M[-2]) / 8192)
            n_ct, = (co // 64,)
            l2 = (8192 * ci * 2 * n_ct + ci * co * 2 * (8192 // tm) + (8192 * co * 2 if res else 0)) / 1e6
            print(f"TM={tm:3d}: " + "  ".join(f"M={m}: {t:6.1f}" for m, t in zip(M, row)) + f"   slope {slope:5.2f} us / 8192 rows, L2->CU {l2:5.0f} MB / 8192 rows = {l2 / slope:5.1f} TB/s")
        _lib.call("wu_set_option", 12, 0)


if __name__ == "__main__":
    main()
