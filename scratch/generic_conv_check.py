"""The generic 3x3 MFMA conv (conv3x3_mfma_kernel) on the shapes that still run on it -- the estimator's 16x16 / 8x8 layers, stride 2,
masked / gated forms, fp32 -- timed inside a captured graph, outputs saved for a bitwise comparison between two libraries:
    python scratch/generic_conv_check.py out_new.pt ; WU_AB_LIB=scratch/_oldlib/libwu_old.so python scratch/generic_conv_check.py out_old.pt
    python scratch/generic_conv_check.py --compare out_new.pt out_old.pt"""
import os
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    if sys.argv[1] == "--compare":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        print(f"{len(a)} outputs compared, {len(bad)} differ: {bad}")
        sys.exit(1 if bad else 0)
    from wu import kernels as K, _lib
    from wu.layout import empty_nhwc, as_nhwc
    dev = torch.device("cuda:0")

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

    outs = {}
    torch.manual_seed(5)
    # (name, B, Cin, Cout, H, W, stride, dtype code, gated)
    cases = [("est 256 @16 B32", 32, 256, 256, 16, 16, 1, _lib.BF16, False), ("est 256 @16 B64", 64, 256, 256, 16, 16, 1, _lib.BF16, False),
             ("est 512 @8 B32", 32, 512, 512, 8, 8, 1, _lib.BF16, False), ("est 512 @8 B64", 64, 512, 512, 8, 8, 1, _lib.BF16, False),
             ("est dgrad 256 @16 B32 gated", 32, 256, 256, 16, 16, 1, _lib.BF16, True),
             ("est s2 128 @64->32 B32", 32, 128, 128, 64, 64, 2, _lib.BF16, False), ("est s2 256 @32->16 B32", 32, 256, 256, 32, 32, 2, _lib.BF16, False),
             ("disc s2 64->128 @128->64 B32", 32, 64, 128, 128, 128, 2, _lib.BF16, False),
             ("fp32 64 @20x24 B2", 2, 64, 64, 20, 24, 1, _lib.F32, True), ("bf16 ragged 64->128 @13x9 B3", 3, 64, 128, 13, 9, 1, _lib.BF16, True)]
    for name, B, ci, co, h, w, st, code, gated in cases:
        tdt = torch.bfloat16 if code == _lib.BF16 else torch.float32
        x = as_nhwc((torch.rand((B, ci, h, w), device=dev) - 0.3), code)
        wt = (torch.rand((co, ci, 3, 3), device=dev) - 0.5) * 0.05
        wf, _ = K.pack_conv3x3(wt, code)
        b = torch.rand(co, device=dev) - 0.5
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        y = empty_nhwc(B, co, ho, wo, tdt, dev)
        eg = as_nhwc(torch.rand((B, co, ho, wo), device=dev) - 0.5, code) if gated else None
        fn = lambda: K.conv3x3(x, wf, b, y, st, K.ACT_RELU, egate=eg, egate_act=K.ACT_RELU if gated else K.ACT_NONE)
        t = graph_time(fn)
        fl = 2.0 * B * ho * wo * 9 * ci * co
        print(f"{name:34s} {t:8.1f} us {fl / t / 1e6:7.0f} TFLOP/s")
        outs[name] = y.clone().cpu()
    torch.save(outs, sys.argv[1])


if __name__ == "__main__":
    main()
