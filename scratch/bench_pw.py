"""Stand-alone timing of the estimator's pointwise-conv GEMM shapes (B=32, 256x256 input): us and TFLOP/s per shape.
    python scratch/bench_pw.py [batch]"""
import os
import statistics
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import resnet as RN, kernels as K
    from wu.layout import empty_nhwc
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    bf = torch.bfloat16

    def run(fn, reps=9):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):                 # back-to-back: a single launch's event pair reads a ~10 us floor
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 8)
        return statistics.median(ts)

    tot = 0.0
    for (s, ci, co, cnt, res) in [(64, 64, 64, 1, False), (64, 64, 256, 4, True), (64, 256, 64, 2, False),
                                  (32, 256, 128, 1, False), (32, 128, 512, 4, True), (32, 512, 128, 3, False),
                                  (16, 512, 256, 1, False), (16, 256, 1024, 23, True), (16, 1024, 256, 22, False),
                                  (8, 1024, 512, 1, False), (8, 512, 2048, 3, True), (8, 2048, 512, 2, False)]:
        x = (torch.rand((B, s, s, ci), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2)
        w = ((torch.rand((co, ci), device=dev) - 0.5) * 0.1).to(bf)
        b = torch.zeros(co, device=dev)
        y = empty_nhwc(B, co, s, s, bf, dev)
        r = (torch.rand((B, s, s, co), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2) if res else None
        t = run(lambda: RN.conv1x1(x, w, b, y, 1, residual=r))
        fl = 2.0 * B * s * s * ci * co
        tot += t * cnt
        print(f"1x1 {ci:4d}->{co:4d} @{s:2d}x{s:<2d} M={B * s * s:6d} {'+res' if res else '    '}: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s   x{cnt} per forward")
    print(f"pointwise convs per estimator forward: {tot / 1e3:.2f} ms")
    for (s, c) in [(64, 64), (32, 128), (16, 256), (8, 512)]:
        x = (torch.rand((B, s, s, c), device=dev) - 0.5).to(bf).permute(0, 3, 1, 2)
        w = (torch.rand((c, c, 3, 3), device=dev) - 0.5) * 0.05
        wf, wd = K.pack_conv3x3(w, 1)
        b = torch.zeros(c, device=dev)
        y = empty_nhwc(B, c, s, s, bf, dev)
        t = run(lambda: K.conv3x3(x, wf, b, y, 1, 1))
        fl = 2.0 * B * s * s * 9 * c * c
        print(f"3x3 {c:4d}->{c:4d} @{s:2d}x{s:<2d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")


if __name__ == "__main__":
    main()
