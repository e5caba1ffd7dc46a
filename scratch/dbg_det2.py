"""Full-size check that the gate-bit producing forward kernels write the same activations as their plain forms (the register-
prefetch bug of the 3->64 kernel was found with this; tests/test_gpu_round2.py::test_gate_bit_producers_at_full_size is its test form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch
from wu import kernels as K
from wu.layout import empty_nhwc, precision_code
dev = torch.device("cuda"); code = precision_code("bf16"); n = 32
torch.manual_seed(0)
# c3
x = torch.rand(n, 3, 256, 256, device=dev) * 2 - 1
w = (torch.rand(64, 3, 3, 3, device=dev) * 2 - 1) * 0.3; b = torch.rand(64, device=dev) - 0.5
y0 = empty_nhwc(n, 64, 256, 256, torch.bfloat16, dev); K.conv3x3_c3(x, w, b, y0, 1, 1, False, code)
for r in range(3):
    y1 = empty_nhwc(n, 64, 256, 256, torch.bfloat16, dev); bits = K.gate_bits_alloc(y1)
    K.conv3x3_c3_bits(x, w, b, y1, bits, 1, code); torch.cuda.synchronize()
    d = (y0 != y1)
    print("c3 bits run", r, "differs from plain in", d.sum().item(), "elements")
for (ci, co, h) in [(64, 128, 128), (128, 256, 64), (256, 512, 32), (768, 256, 64), (384, 128, 128), (192, 64, 256)]:
    xx = empty_nhwc(n, ci, h, h, torch.bfloat16, dev); xx.copy_(torch.rand(n, ci, h, h, device=dev) * 2 - 1)
    ww = (torch.rand(co, ci, 3, 3, device=dev) * 2 - 1) * 0.05
    wf, wd = K.pack_conv3x3(ww, code); bb = torch.rand(co, device=dev) - 0.5
    y0 = empty_nhwc(n, co, h, h, torch.bfloat16, dev); K.conv3x3(xx, wf, bb, y0, 1, 1)
    res = []
    for r in range(3):
        y1 = empty_nhwc(n, co, h, h, torch.bfloat16, dev); bits = K.gate_bits_alloc(y1)
        K.conv3x3_bits(xx, wf, bb, y1, 1, gate_bits_out=bits); torch.cuda.synchronize()
        res.append((y0 != y1).sum().item())
    print(f"{ci}->{co} @{h}: bits-variant differs from plain in {res}")
