"""do the data-gradient and weight-gradient kernels of one layer finish sooner when each gets its own
compute units (two hipExtStreamCreateWithCUMask streams, persistent grids sized to their share) than back to back on the whole chip?

For every layer: (1) dgrad then wgrad on ONE stream, whole chip; (2) the product's arrangement: two ordinary streams, both grids
= 256 workgroups; (3) CU-masked streams, A CUs for the data gradient / 256 - A for the weight gradient, grids A / 256 - A.
Wall time from a common start event to the later of the two end events, median of 7 rounds, interleaved on one box."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
B = 32
OPT_GRID_CUS = 10


def masked_stream(bits):
    words = (ctypes.c_uint * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    out = ctypes.c_void_p()
    _lib.call("wu_stream_create_cu_mask", words, 8, ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value, device=DEV)


def main():
    layers = [("u1.2", 64, 64, 256), ("u2.0", 384, 128, 128), ("d3.2", 256, 256, 64), ("u3.0", 768, 256, 64)]
    splits = [(128, 128), (144, 112), (160, 96)]
    streams = {}
    for a, b in splits:
        # bit i of the mask = logical CU i; KFD deals logical CUs round-robin over the XCDs, so a prefix of the bit string takes
        # the same number of CUs from every XCD
        streams[(a, b)] = (masked_stream(set(range(a))), masked_stream(set(range(a, 256))))
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    print(f"{'layer':6s} {'dgrad':>8s} {'wgrad':>8s} {'serial':>8s} {'2 streams':>10s} " + " ".join(f"{f'mask {a}/{b}':>13s}" for a, b in splits))
    for name, ci, co, s in layers:
        # forward conv ci -> co at s x s;  data gradient: gy (co ch) -> dx (ci ch);  weight gradient: x (ci ch), gy (co ch)
        x = (torch.rand((B, s, s, ci), device=DEV) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        gy = (torch.rand((B, s, s, co), device=DEV) * 2 - 1).to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.rand((co, ci, 3, 3), device=DEV) - 0.5) * 0.1
        _, w_dgrad = K.pack_conv3x3(w, _lib.BF16)
        dx = torch.empty((B, s, s, ci), device=DEV, dtype=torch.bfloat16).permute(0, 3, 1, 2)
        dw = torch.empty((co, ci, 3, 3), device=DEV)
        db = torch.empty((co,), device=DEV)

        def dgrad(cus=0):
            _lib.call("wu_set_option", OPT_GRID_CUS, cus)
            K.conv3x3(gy, w_dgrad, None, dx)

        def wgrad(cus=0):
            _lib.call("wu_set_option", OPT_GRID_CUS, cus)
            K.conv3x3_wgrad(x, gy, dw, db)

        def timed(fa, fb, sa, sb, reps=3):
            """fa on stream sa, fb on stream sb, `reps` pairs; us per pair."""
            cur = torch.cuda.current_stream(DEV)
            e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(cur)
            sa.wait_event(e0)
            sb.wait_event(e0)
            for _ in range(reps):
                with torch.cuda.stream(sa):
                    fa()
                with torch.cuda.stream(sb):
                    fb()
            ea.record(sa)
            eb.record(sb)
            torch.cuda.synchronize()
            return max(e0.elapsed_time(ea), e0.elapsed_time(eb)) / reps * 1e3

        cur = torch.cuda.current_stream(DEV)
        for _ in range(2):
            dgrad(); wgrad()
        torch.cuda.synchronize()
        ref = (dx.clone(), dw.clone())
        res = {k: [] for k in ["d", "w", "serial", "two"] + splits}
        for _ in range(7):
            res["d"].append(timed(dgrad, lambda: None, cur, cur))
            res["w"].append(timed(wgrad, lambda: None, cur, cur))
            res["serial"].append(timed(dgrad, wgrad, cur, cur))
            res["two"].append(timed(dgrad, wgrad, s1, s2))
            for a, b in splits:
                sa, sb = streams[(a, b)]
                res[(a, b)].append(timed(lambda: dgrad(a), lambda: wgrad(b), sa, sb))
        _lib.call("wu_set_option", OPT_GRID_CUS, 0)
        assert torch.equal(dx, ref[0]), "masked-stream data gradient differs"
        rel = ((dw - ref[1]).norm() / ref[1].norm()).item()      # the split-K factor follows the grid: another fp32 summation order
        med = {k: statistics.median(v) for k, v in res.items()}
        print(f"{name:6s} {med['d']:8.1f} {med['w']:8.1f} {med['serial']:8.1f} {med['two']:10.1f} " +
              " ".join(f"{med[k]:13.1f}" for k in splits) + f"   (dW rel diff {rel:.1e})")


if __name__ == "__main__":
    main()
