

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0'); B = 32
    c, h, cs = 128, 128, 64
    x = (torch.rand((B, h, h, c), device=dev)).to(torch.bfloat16).permute(0, 3, 1, 2)
    cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, torch.bfloat16, dev)
    g = (torch.rand((B, 2 * h, 2 * h, c + cs), device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    ys = torch.rand((B, c), device=dev) + 0.5; ym = torch.rand((B, c), device=dev)
    st = K.adain_stats(x, 1e-5)
    dx = empty_nhwc(B, c, h, h, torch.bfloat16, dev)
    for _ in range(2):
        mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
        K.adain_upcat_bwd(g, x, st, ys, dx, 0.3, 123, mb, 1)
    xi = torch.rand((B, 3, 256, 256), device=dev) * 2 - 1
    w = torch.rand((64, 3, 3, 3), device=dev) - 0.5; b = torch.zeros(64, device=dev)
    y = empty_nhwc(B, 64, 256, 256, torch.bfloat16, dev)
    for _ in range(2): K.conv3x3_c3(xi, w, b, y, 1, 1, False, 1)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
