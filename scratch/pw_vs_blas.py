"""What does the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reach on the estimator's pointwise shapes?  A yardstick for
conv1x1_mfma_kernel, not a product path.  Both are timed as a captured graph of 20 launches (no host time in the figure).
    python scratch/pw_vs_blas.py"""
import os
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import resnet as RN
    from wu.layout import empty_nhwc
    dev = torch.device("cuda:0")
    bf = torch.bfloat16

    def graph_time(fn, n=20, reps=7):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        return sorted(ts)[len(ts) // 2]

    from wu import _lib
    print(f"{'shape':30s} {'64x64 us':>8s} {'TF/s':>5s} | {'4w D2':>6s} {'4w D4':>6s} {'8w D2':>6s} {'8w D3':>6s} {'8w D4':>6s} {'best TF/s':>9s} | {'mm us':>7s} {'TF/s':>5s}")
    for (ci, co, res) in [(256, 1024, True), (1024, 256, False), (512, 2048, True), (2048, 512, False), (128, 512, True), (512, 128, False),
                          (64, 256, True), (256, 64, False)]:
        for M in (8192, 16384, 32768):
            B, s = M // 256, 16
            x = (torch.rand((B, s, s, ci), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2)
            w = ((torch.rand((co, ci), device=dev) - 0.5) * 0.1).to(bf)
            b = torch.zeros(co, device=dev)
            y = empty_nhwc(B, co, s, s, bf, dev)
            r = (torch.rand((B, s, s, co), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2) if res else None
            _lib.call("wu_set_option", 15, 0)
            t_wu = graph_time(lambda: RN.conv1x1(x, w, b, y, 1, residual=r))
            t3 = []
            for d in (2, 4, 8 + 2, 8 + 3, 8 + 4):
                _lib.call("wu_set_option", 15, d)
                t3.append(graph_time(lambda: RN.conv1x1(x, w, b, y, 1, residual=r)) if co % 128 == 0 else float("nan"))
            _lib.call("wu_set_option", 15, 0)
            x2 = x.permute(0, 2, 3, 1).reshape(M, ci)
            wt = w.t()
            out = torch.empty((M, co), device=dev, dtype=bf)
            t_mm = graph_time(lambda: torch.mm(x2, wt, out=out))
            fl = 2.0 * M * ci * co
            print(f"{ci:4d}->{co:4d} M={M:6d} {'+res' if res else '    '}        {t_wu:8.1f} {fl / t_wu / 1e6:5.0f} | {t3[0]:6.1f} {t3[1]:6.1f} {t3[2]:6.1f} {t3[3]:6.1f} {t3[4]:6.1f} {fl / min(t3) / 1e6:9.0f} | {t_mm:7.1f} {fl / t_mm / 1e6:5.0f}")


if __name__ == "__main__":
    main()
