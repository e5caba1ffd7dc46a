

def main():
    import sys, os, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0'); B = 32
    def run(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for (c, h, cs) in [(128, 128, 64), (256, 64, 128), (512, 32, 256)]:
        x = (torch.rand((B, h, h, c), device=dev)).to(torch.bfloat16).permute(0, 3, 1, 2)
        cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, torch.bfloat16, dev)
        g = (torch.rand((B, 2 * h, 2 * h, c + cs), device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        ys = torch.rand((B, c), device=dev) + 0.5; ym = torch.rand((B, c), device=dev)
        st = K.adain_stats(x, 1e-5)
        dx = empty_nhwc(B, c, h, h, torch.bfloat16, dev)
        out_mb = B * 4 * h * h * c * 2 / 1e6
        for p in (0.0, 0.3):
            t_f = run(lambda: K.adain_upcat(x, st, ys, ym, cat, p, 123, True))
            mb = K.adain_upcat(x, st, ys, ym, cat, p, 123, True)
            t_b = run(lambda: K.adain_upcat_bwd(g, x, st, ys, dx, p, 123, mb, 1))
            print(f"C={c:3d} {h}->{2*h}: p={p}: fwd {t_f:7.1f} us ({out_mb/t_f:.2f} TB/s of output)   bwd(gather+apply) {t_b:7.1f} us ({out_mb/t_b:.2f} TB/s of dy)")
        t_s = run(lambda: K.adain_stats(x, 1e-5))
        print(f"   stats {t_s:.1f} us")
    # maxpool
    for (c, h) in [(64, 256), (128, 128), (256, 64)]:
        x = (torch.rand((B, h, h, c), device=dev)).to(torch.bfloat16).permute(0, 3, 1, 2)
        y = empty_nhwc(B, c, h // 2, h // 2, torch.bfloat16, dev); gy = y.clone(); dxx = empty_nhwc(B, c, h, h, torch.bfloat16, dev); gs = x.clone()
        mbs = B * h * h * c * 2 / 1e6
        print(f"maxpool C={c} {h}: fwd {run(lambda: K.maxpool2(x, y)):.1f} us  bwd(+skip,+gate) {run(lambda: K.maxpool2_bwd(x, gy, dxx, gs, 1)):.1f} us  (tensor {mbs:.0f} MB)")
    # c3 fwd, conv1x1
    xi = torch.rand((B, 3, 256, 256), device=dev) * 2 - 1
    w = torch.rand((64, 3, 3, 3), device=dev) - 0.5; b = torch.zeros(64, device=dev)
    y = empty_nhwc(B, 64, 256, 256, torch.bfloat16, dev)
    print(f"c3 fwd: {run(lambda: K.conv3x3_c3(xi, w, b, y, 1, 1, False, 1)):.1f} us (output 268 MB)")
    gy = (torch.rand((B, 256, 256, 64), device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    dw = torch.zeros((64, 3, 3, 3), device=dev); db = torch.zeros(64, device=dev)
    print(f"c3 wgrad: {run(lambda: K.conv3x3_c3_wgrad(xi, gy, dw, db, 1, 1)):.1f} us")
    w3 = torch.rand((3, 64), device=dev) - 0.5; b3 = torch.zeros(3, device=dev); out = torch.empty((B, 3, 256, 256), device=dev)
    print(f"conv1x1 fwd: {run(lambda: K.conv1x1_tanh(y, w3, b3, out)):.1f} us")
    dxx = empty_nhwc(B, 64, 256, 256, torch.bfloat16, dev); dw3 = torch.zeros((3, 64), device=dev); db3 = torch.zeros(3, device=dev)
    print(f"conv1x1 bwd: {run(lambda: K.conv1x1_tanh_bwd(out, out, y, w3, dxx, dw3, db3, 1)):.1f} us")


if __name__ == "__main__":
    main()
