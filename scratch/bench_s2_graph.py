"""Stride-2 3x3 convs (SNDisc trunk, estimator's first block of layers 2-4) forward: register-staged kernel against the gathered-row pipeline,
launches timed inside a captured graph.   python scratch/bench_s2_graph.py [batch]"""
import os
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import kernels as K, _lib
    from wu.layout import empty_nhwc, as_nhwc
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32

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

    tot = [0.0, 0.0]
    totd = [0.0, 0.0]
    for name, ci, co, h in [("disc 64->128 @128", 64, 128, 128), ("disc 128->256 @64", 128, 256, 64), ("disc 256->512 @32", 256, 512, 32),
                            ("est 128->128 @64", 128, 128, 64), ("est 256->256 @32", 256, 256, 32), ("est 512->512 @16", 512, 512, 16)]:
        for bb in (B, 2 * B):
            x = as_nhwc(torch.rand((bb, ci, h, h), device=dev) - 0.3, _lib.BF16)
            wt = (torch.rand((co, ci, 3, 3), device=dev) - 0.5) * 0.05
            wf, _ = K.pack_conv3x3(wt, _lib.BF16)
            b = torch.rand(co, device=dev) - 0.5
            y = empty_nhwc(bb, co, h // 2, h // 2, torch.bfloat16, dev)
            ts = []
            for opt in (0, 2 + 8):
                _lib.call("wu_set_option", 15, opt)
                ts.append(graph_time(lambda: K.conv3x3(x, wf, b, y, 2, K.ACT_LEAKY)))
            _lib.call("wu_set_option", 15, 2 + 8 + (128 << 5))
            fl = 2.0 * bb * (h // 2) ** 2 * 9 * ci * co
            if bb == B:
                tot[0] += ts[0]; tot[1] += ts[1]
            # data gradient: dX (bb, ci, h, h) from dY (bb, co, h/2, h/2), LeakyReLU gate of the consumer in the epilogue
            _, wd = K.pack_conv3x3(wt, _lib.BF16)
            gy = as_nhwc(torch.rand((bb, co, h // 2, h // 2), device=dev) - 0.5, _lib.BF16)
            dx = empty_nhwc(bb, ci, h, h, torch.bfloat16, dev)
            td = []
            for opt in (0, 2 + 8):
                _lib.call("wu_set_option", 15, opt)
                td.append(graph_time(lambda: K.conv3x3_s2_dgrad(gy, wd, dx, egate=x, egate_act=K.ACT_LEAKY)))
            _lib.call("wu_set_option", 15, 2 + 8 + (128 << 5))
            if bb == B:
                totd[0] += td[0]; totd[1] += td[1]
            print(f"{name:20s} B={bb:4d}: fwd {ts[0]:7.1f} us {fl / ts[0] / 1e6:5.0f} TF/s -> {ts[1]:7.1f} us {fl / ts[1] / 1e6:5.0f} TF/s | dgrad {td[0]:7.1f} us {fl / td[0] / 1e6:5.0f} TF/s -> {td[1]:7.1f} us {fl / td[1] / 1e6:5.0f} TF/s")
    print(f"sum at B={B}: fwd {tot[0]:.1f} -> {tot[1]:.1f} us, dgrad {totd[0]:.1f} -> {totd[1]:.1f} us   (left: register-staged kernels, right: gathered rows on the LDS-DMA GEMM pipeline)")


if __name__ == "__main__":
    main()
