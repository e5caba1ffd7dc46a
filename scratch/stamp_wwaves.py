

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    dev = torch.device('cuda:0'); B = 32
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    for name, ci, co, s in [('d4.2', 512, 512, 32), ('u3.0', 768, 256, 64)]:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        gy = (torch.rand((B, s, s, co), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        dw = torch.empty((co, ci, 3, 3), device=dev)
        for _ in range(5): K.conv3x3_wgrad(x, gy, dw, None)
        torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', dbg.data_ptr()); dbg.zero_()
        K.conv3x3_wgrad(x, gy, dw, None); torch.cuda.synchronize()
        _lib.call('wu_set_debug_buffer', None)
        d = dbg.view(256, 8, 8).double().cpu()
        d = d[d[:, 0, 6] > 0]
        tiles = d[:, 0, 6].mean().item()
        per = d.mean(dim=0) / tiles
        print(f"{name} (no bias): per tile, waves 0..7: compute " + " ".join(f"{v:.0f}" for v in per[:, 1].tolist()) + " | barrier " + " ".join(f"{v:.0f}" for v in per[:, 4].tolist()) + " | dma wait " + " ".join(f"{v:.0f}" for v in per[:, 0].tolist()))


if __name__ == "__main__":
    main()
