"""A/B: gated data-gradient conv with the gate as a tensor vs as bits (same kernel family, same box).
    python scratch/ab_gatebits.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch
from wu import kernels as K
from wu.layout import empty_nhwc, precision_code


def bench(fn, reps=9):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


dev = torch.device("cuda")
code = precision_code("bf16")
from wu import _lib
if len(sys.argv) > 1:
    _lib.call("wu_set_option", 6, int(sys.argv[1]))
for (n, c, h) in [(32, 64, 256), (32, 128, 128), (32, 256, 64)]:
    torch.manual_seed(0)
    gy = empty_nhwc(n, c, h, h, torch.bfloat16, dev); gy.copy_(torch.randn(n, c, h, h, device=dev))
    mid = empty_nhwc(n, c, h, h, torch.bfloat16, dev); mid.copy_(torch.relu(torch.randn(n, c, h, h, device=dev)))
    w = (torch.randn(c, c, 3, 3, device=dev) * 0.05)
    wf, wd = K.pack_conv3x3(w, code)
    out = empty_nhwc(n, c, h, h, torch.bfloat16, dev)
    bits = K.gate_bits_alloc(mid)
    b = torch.zeros(c, device=dev)
    # bits of `mid` via a forward that writes them (content irrelevant for timing, but make them real: conv of something)
    K.conv3x3_bits(gy, wf, b, out, 1, gate_bits_out=bits)
    t_fwd = bench(lambda: K.conv3x3(gy, wf, b, out, 1, 1))
    t_fwdb = bench(lambda: K.conv3x3_bits(gy, wf, b, out, 1, gate_bits_out=bits))
    t_t = bench(lambda: K.conv3x3(gy, wd, None, out, 1, 0, egate=mid, egate_act=1))
    t_b = bench(lambda: K.conv3x3_bits(gy, wd, None, out, 0, egate_bits=bits))
    t_u = bench(lambda: K.conv3x3(gy, wd, None, out, 1, 0))
    print(f"{c}->{c} @{h} B={n}: fwd {t_fwd:.1f} us, fwd+bits {t_fwdb:.1f} us | dgrad ungated {t_u:.1f}, tensor gate {t_t:.1f}, bit gate {t_b:.1f} us")
