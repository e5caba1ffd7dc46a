

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    dev = torch.device('cuda:0'); B = 32
    NWAVES = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    _lib.call('wu_set_option', 2, 1 if NWAVES == 8 else 2)
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    for name, ci, co, s in [('d1.2', 64, 64, 256), ('d2.2', 128, 128, 128), ('d4.2', 512, 512, 32), ('u1.0', 192, 64, 256), ('u3.0', 768, 256, 64)]:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        gy = (torch.rand((B, s, s, co), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        dw = torch.empty((co, ci, 3, 3), device=dev); db = torch.empty((co,), device=dev)
        for _ in range(20): K.conv3x3_wgrad(x, gy, dw, db)
        torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', dbg.data_ptr()); dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.conv3x3_wgrad(x, gy, dw, db); e1.record(); torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', None)
        d = dbg.view(256, 8, 8).double().cpu()[:, :NWAVES]
        d = d[d[:, 0, 6] > 0]
        tiles = d[:, 0, 6].mean().item()
        clk = (d[:, 0, 2] / d[:, 0, 3]).median().item() * 0.1
        w, c, tot, bar = d[:, :, 0].mean().item(), d[:, :, 1].mean().item(), d[:, :, 2].mean().item(), d[:, :, 4].mean().item()
        print(f"{name}: wgrad+reduce {e0.elapsed_time(e1)*1e3:.0f} us, tiles/WG {tiles:.1f}; per tile: dma wait {w/tiles:.0f}, barrier {bar/tiles:.0f}, compute {c/tiles:.0f} (ideal 4608); main loop total/tile {tot/tiles:.0f}; in-kernel clock {clk:.2f} GHz")


if __name__ == "__main__":
    main()
