"""In-process A/B of conv kernel variants (same device, interleaved rounds): per-layer median times."""


def main():
    import sys, os, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0')
    B = 32
    layers = [('d1.2', 64, 64, 256), ('d2.2', 128, 128, 128), ('d3.2', 256, 256, 64), ('d4.2', 512, 512, 32), ('u3.0', 768, 256, 64), ('u2.0', 384, 128, 128), ('u1.0', 192, 64, 256), ('dg u1.0', 64, 192, 256)]
    variants = {'v1': (0, 0), 'v2 8 waves': (3, 1), 'v2 4 waves': (2, 1)}
    def run(fn, reps=3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps): fn()
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e3
    for name, ci, co, s in layers:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        w = ((torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05)
        wf, wd = K.pack_conv3x3(w, 1)
        bias = torch.zeros(co, device=dev)
        y = empty_nhwc(B, co, s, s, torch.bfloat16, dev)
        gf = 2 * B * s * s * 9 * ci * co / 1e9
        res = {k: [] for k in variants}
        outs = {}
        for rnd in range(7):
            for k, (v2, pers) in variants.items():
                _lib.call('wu_set_option', 0, v2); _lib.call('wu_set_option', 1, pers)
                t = run(lambda: K.conv3x3(x, wf, bias, y, 1, 1))
                res[k].append(t)
                if rnd == 0: outs[k] = y.clone()
        same = all(torch.equal(outs['v1'], o) for o in outs.values())
        print(f"{name:8s} {ci:4d}->{co:4d} @{s:3d}: " + "  ".join(f"{k}: {statistics.median(v):7.1f} us ({gf/statistics.median(v)*1e3/1e3:5.0f} TF)" for k, v in res.items()) + f"   bitwise-equal: {same}")
    _lib.call('wu_set_option', 0, 1); _lib.call('wu_set_option', 1, 1)


if __name__ == "__main__":
    main()
