

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0'); B = 32
    NWAVES = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    BITS = len(sys.argv) > 3 and sys.argv[3] == 'bits'        # forward with gate-bit output
    if len(sys.argv) > 2: _lib.call('wu_set_option', 6, int(sys.argv[2]))
    _lib.call('wu_set_option', 0, 3 if NWAVES == 8 else 2)
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    for name, ci, co, s in [('d1.2', 64, 64, 256), ('d2.2', 128, 128, 128), ('d4.2', 512, 512, 32), ('u1.0', 192, 64, 256)]:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        w = ((torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05)
        wf, wd = K.pack_conv3x3(w, 1); bias = torch.zeros(co, device=dev)
        y = empty_nhwc(B, co, s, s, torch.bfloat16, dev)
        gb = K.gate_bits_alloc(y)
        run = (lambda: K.conv3x3_bits(x, wf, bias, y, 1, gate_bits_out=gb)) if BITS else (lambda: K.conv3x3(x, wf, bias, y, 1, 1))
        for _ in range(3): run()
        torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', dbg.data_ptr()); dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', None)
        d = dbg.view(256, 8, 8).double().cpu()[:, :NWAVES]
        raw7 = dbg.view(256, 8, 8)[:, :NWAVES, 7].cpu()
        tiles, chunks = d[0, 0, 6].item(), float(int(raw7[0, 0].item()) & 255)
        first = (raw7 >> 8).double().mean().item()
        clk = (d[:, 0, 2] / d[:, 0, 3]).median().item() * 0.1
        ph = d[:, :, :6].mean(dim=(0, 1)); ph[2] = 0; ph[3] = 0
        tot = ph.sum().item()
        names = ['dma wait', 'compute', '-', '-', 'chunk-top barrier', 'epilogue']
        print(f"{name}: kernel {e0.elapsed_time(e1)*1e3:.0f} us, tiles/WG {tiles:.0f}, chunks {chunks:.0f}; cycles per tile: " +
              ", ".join(f"{n} {v/tiles:.0f}" for n, v in zip(names, ph.tolist())) + f"; in-kernel clock {clk:.2f} GHz; first chunk {first/tiles:.0f}, later chunks avg {(ph[1].item() - first)/tiles/max(chunks-1,1):.0f}; total/tile {tot/tiles:.0f} cyc; compute/chunk {ph[1].item()/tiles/chunks:.0f}")


if __name__ == "__main__":
    main()
