"""Interleaved A/B of ONE kernel-variant switch (wu_set_option key: values 0 / 1) on the 64-channel 256x256 conv launches.
    python scratch/ab_opt.py <key> [rounds]"""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc


def main():
    key = int(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev, bf, B, s, c = torch.device("cuda:0"), torch.bfloat16, 32, 256, 64
    x = (torch.rand((B, s, s, c), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
    w = (torch.rand((c, c, 3, 3), device=dev) * 2 - 1) * 0.05
    wf, wd = K.pack_conv3x3(w, 1)
    bias = torch.rand(c, device=dev) - 0.5
    y, pl = empty_nhwc(B, c, s, s, bf, dev), empty_nhwc(B, c, s // 2, s // 2, bf, dev)
    gb = torch.randint(-2**31, 2**31 - 1, (_lib.load().wu_gate_bits_bytes(B, s, s, c) // 4,), dtype=torch.int32, device=dev)
    cases = {"fwd+pool (down1.2)": lambda: K.conv3x3_relu_pool(x, wf, bias, y, pl), "fwd (up1.2)": lambda: K.conv3x3(x, wf, bias, y, 1, 1),
             "dgrad bits": lambda: K.conv3x3_bits(x, wd, None, y, egate_bits=gb)}
    if key == 12:      # (halo residency experiment, round 3: not built -- kept for the record of what was timed)
        # several cout tiles: the data gradient of up1.0 (64 -> 192 @256) and the forward of down2.0 (64 -> 128 @128, gate bits)
        w3 = (torch.rand((64, 192, 3, 3), device=dev) * 2 - 1) * 0.05            # forward 192 -> 64: its dgrad pack maps 64 -> 192
        _, wd3 = K.pack_conv3x3(w3, 1)
        y = empty_nhwc(B, 192, s, s, bf, dev)
        x2 = (torch.rand((B, 128, 128, 64), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
        w2 = (torch.rand((128, 64, 3, 3), device=dev) * 2 - 1) * 0.05
        wf2, _ = K.pack_conv3x3(w2, 1)
        b2 = torch.rand(128, device=dev) - 0.5
        y2 = empty_nhwc(B, 128, 128, 128, bf, dev)
        gb2 = K.gate_bits_alloc(y2)
        cases = {"dgrad up1.0 64->192": lambda: K.conv3x3(x, wd3, None, y), "fwd down2.0 64->128": lambda: K.conv3x3_bits(x2, wf2, b2, y2, 1, gate_bits_out=gb2)}
        ys = {"dgrad up1.0 64->192": y, "fwd down2.0 64->128": y2}

    def timed(fn, reps=4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for name, fn in cases.items():
        outs, res = {}, {0: [], 1: []}
        for v in (0, 1):
            _lib.call("wu_set_option", key, v); fn(); torch.cuda.synchronize(); outs[v] = (ys[name] if key == 12 else y).clone()
        for _ in range(rounds):
            for v in (0, 1):
                _lib.call("wu_set_option", key, v); res[v].append(timed(fn))
        m0, m1 = statistics.median(res[0]), statistics.median(res[1])
        print(f"{name:22s} option {key}: 0 -> {m0:7.1f} us   1 -> {m1:7.1f} us   ratio {m1 / m0:.3f}   bitwise equal {torch.equal(outs[0], outs[1])}")
    _lib.call("wu_set_option", key, 1)


if __name__ == "__main__":
    main()
