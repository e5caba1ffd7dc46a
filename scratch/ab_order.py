

def main():
    import sys, os, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    from wu.layout import empty_nhwc
    dev = torch.device('cuda:0'); B = 32
    layers = [('d2.2', 128, 128, 128), ('d3.2', 256, 256, 64), ('d4.2', 512, 512, 32), ('u3.0', 768, 256, 64), ('u2.0', 384, 128, 128), ('dg u1.0', 64, 192, 256), ('dg u2.0', 128, 384, 128)]
    def run(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for name, ci, co, s in layers:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        w = ((torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05)
        wf, wd = K.pack_conv3x3(w, 1); bias = torch.zeros(co, device=dev)
        y = empty_nhwc(B, co, s, s, torch.bfloat16, dev)
        gf = 2 * B * s * s * 9 * ci * co / 1e9
        res = {0: [], 1: []}; outs = {}
        for rnd in range(7):
            for order in (0, 1):
                _lib.call('wu_set_option', 7, order)
                res[order].append(run(lambda: K.conv3x3(x, wf, bias, y, 1, 1)))
                if rnd == 0: outs[order] = y.clone()
        print(f"{name:8s} {ci:4d}->{co:4d} @{s:3d}: contiguous {statistics.median(res[0]):7.1f} us ({gf/statistics.median(res[0]):.3f} PF)   strided {statistics.median(res[1]):7.1f} us ({gf/statistics.median(res[1]):.3f} PF)  equal {torch.equal(outs[0], outs[1])}")
    _lib.call('wu_set_option', 7, 0)


if __name__ == "__main__":
    main()
