

def main():
    import sys, os, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu import _lib, kernels as K
    dev = torch.device('cuda:0'); B = 32
    layers = [('d1.2', 64, 64, 256), ('d2.2', 128, 128, 128), ('d3.2', 256, 256, 64), ('d4.2', 512, 512, 32), ('u3.0', 768, 256, 64), ('u2.0', 384, 128, 128), ('u1.0', 192, 64, 256)]
    def run(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for name, ci, co, s in layers:
        x = (torch.rand((B, s, s, ci), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        gy = (torch.rand((B, s, s, co), device=dev) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        dw = torch.empty((co, ci, 3, 3), device=dev); db = torch.empty((co,), device=dev)
        gf = 2 * B * s * s * 9 * ci * co / 1e9
        res = {0: [], 1: []}; outs = {}
        for rnd in range(7):
            for v in (0, 1):
                _lib.call('wu_set_option', 2, v + 1)
                res[v].append(run(lambda: K.conv3x3_wgrad(x, gy, dw, db)))
                if rnd == 0: outs[v] = (dw.clone(), db.clone())
        rel = ((outs[0][0] - outs[1][0]).norm() / outs[0][0].norm()).item()
        print(f"{name:6s} {ci:4d}->{co:4d} @{s:3d}: 8 waves {statistics.median(res[0]):7.1f} us ({gf/statistics.median(res[0]):.3f} PF)   4 waves {statistics.median(res[1]):7.1f} us ({gf/statistics.median(res[1]):.3f} PF)  (incl. reduce)  rel diff {rel:.2e}")
    _lib.call('wu_set_option', 2, 1)


if __name__ == "__main__":
    main()
