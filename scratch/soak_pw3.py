"""Repeatability soak of the persistent LDS-DMA GEMM pipeline (conv1x1_pw3_kernel): every estimator pointwise shape (plain / residual + ReLU / gate / residual +
gate) and the stride-2 forward / data-gradient shapes, REPS launches each at full size, every result compared bit for bit with the first -- alone, and beside a
second stream that keeps the chip busy with the LDS-DMA 3x3 conv (the kernels of two streams share CUs in the GAN step).  An intermittent hazard (a wait that
is one operation short, a register read before its load landed) shows up as a run that differs.
    python scratch/soak_pw3.py [reps]"""
import os
import sys


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
    import torch
    from wu import resnet as RN, kernels as K, _lib
    from wu.layout import empty_nhwc, as_nhwc
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.manual_seed(1)

    # background load: a 64 -> 64 @256^2 conv on a second stream
    side = torch.cuda.Stream()
    bx = as_nhwc(torch.rand((16, 64, 256, 256), device=dev) - 0.3, _lib.BF16)
    bw, _ = K.pack_conv3x3((torch.rand((64, 64, 3, 3), device=dev) - 0.5) * 0.05, _lib.BF16)
    bb = torch.zeros(64, device=dev)
    by = empty_nhwc(16, 64, 256, 256, bf, dev)

    def soak(name, fn, out):
        bad = 0
        for busy in (False, True):
            fn(); torch.cuda.synchronize()
            ref = out.clone()
            for r in range(reps):
                if busy:
                    with torch.cuda.stream(side):
                        K.conv3x3(bx, bw, bb, by, 1, K.ACT_RELU)
                out.fill_(0)
                fn()
                if r % 10 == 9 or r == reps - 1:
                    torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad += 1
            torch.cuda.synchronize()
        print(f"{name:60s} {2 * reps} launches, {bad} differ")
        return bad

    total = 0
    for B in (32, 64):
        for (s, ci, co) in [(64, 64, 256), (64, 256, 64), (32, 128, 512), (32, 512, 128), (16, 256, 1024), (16, 1024, 256), (8, 512, 2048), (8, 2048, 512)]:
            x = as_nhwc(torch.rand((B, ci, s, s), device=dev) - 0.5, _lib.BF16)
            w = ((torch.rand((co, ci), device=dev) - 0.5) * 0.1).to(bf)
            b = torch.rand(co, device=dev) - 0.5
            y = empty_nhwc(B, co, s, s, bf, dev)
            res = as_nhwc(torch.rand((B, co, s, s), device=dev) - 0.5, _lib.BF16)
            gate = as_nhwc(torch.rand((B, co, s, s), device=dev) - 0.5, _lib.BF16)
            for tag, kw in (("plain", dict()), ("res+relu", dict(act=1, residual=res)), ("gate", dict(egate=gate, egate_act=1)), ("res+gate", dict(residual=res, egate=gate, egate_act=1))):
                total += soak(f"1x1 {ci}->{co} @{s} B={B} {tag}", lambda: RN.conv1x1(x, w, b, y, **kw), y)
    for B in (32, 64):
        for (ci, co, h) in [(64, 128, 128), (128, 256, 64), (256, 512, 32), (128, 128, 64), (256, 256, 32), (512, 512, 16)]:
            x = as_nhwc(torch.rand((B, ci, h, h), device=dev) - 0.3, _lib.BF16)
            wt = (torch.rand((co, ci, 3, 3), device=dev) - 0.5) * 0.05
            wf, wd = K.pack_conv3x3(wt, _lib.BF16)
            b = torch.rand(co, device=dev) - 0.5
            y = empty_nhwc(B, co, h // 2, h // 2, bf, dev)
            total += soak(f"3x3 s2 fwd {ci}->{co} @{h} B={B}", lambda: K.conv3x3(x, wf, b, y, 2, K.ACT_LEAKY), y)
            gy = as_nhwc(torch.rand((B, co, h // 2, h // 2), device=dev) - 0.5, _lib.BF16)
            dx = empty_nhwc(B, ci, h, h, bf, dev)
            total += soak(f"3x3 s2 dgrad {ci}->{co} @{h} B={B} (gated)", lambda: K.conv3x3_s2_dgrad(gy, wd, dx, egate=x, egate_act=K.ACT_LEAKY), dx)
    print("TOTAL differing launches:", total)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
