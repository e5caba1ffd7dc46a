

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0'); B = 32
    _lib.call('wu_set_option', 0, 3)
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    for prio in (0, 1):
        _lib.call('wu_set_option', 6, prio)
        for name, ci, co, s in [('d4.2', 512, 512, 32), ('d1.2', 64, 64, 256)]:
            x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
            w = ((torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05)
            wf, wd = K.pack_conv3x3(w, 1); bias = torch.zeros(co, device=dev)
            y = empty_nhwc(B, co, s, s, torch.bfloat16, dev)
            for _ in range(5): K.conv3x3(x, wf, bias, y, 1, 1)
            torch.cuda.synchronize()
            _lib.call('wu_set_debug_buffer', dbg.data_ptr()); dbg.zero_()
            K.conv3x3(x, wf, bias, y, 1, 1); torch.cuda.synchronize()
            _lib.call('wu_set_debug_buffer', None)
            d = dbg.view(256, 8, 8).double().cpu()
            tiles, chunks = d[0, 0, 6].item(), d[0, 0, 7].item()
            per = d[:, :, :].mean(dim=0) / (tiles * chunks)
            print(f"prio={prio} {name}: per chunk, waves 0..7: compute " + " ".join(f"{v:.0f}" for v in per[:, 1].tolist()) + " | barrier " + " ".join(f"{v:.0f}" for v in per[:, 4].tolist()) + " | dma wait " + " ".join(f"{v:.0f}" for v in per[:, 0].tolist()))


if __name__ == "__main__":
    main()
