"""wu_conv1x1_chain against the two wu_conv1x1_fwd launches it replaces, at the estimator's shapes (ResNet-101, 256x256 input): forward form
(bias + residual + ReLU, bias + ReLU) and backward form (residual + gate, gate); 4 back-to-back launches per timing, median of 7."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch
from wu import resnet as RN
from wu.layout import empty_nhwc
dev, bf = torch.device("cuda:0"), torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def run(fn, reps=7, inner=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return statistics.median(ts)
def t(c, s):
    return torch.relu(torch.randn((B, s, s, c), device=dev)).to(bf).permute(0, 3, 1, 2)
RELU, NONE = RN.RELU, RN.NONE
tot = [0.0] * 4
for name, s, k1, c1, c2, count in (("layer1 pair", 64, 64, 256, 64, 2), ("layer1->2", 64, 64, 256, 128, 1), ("layer2 pair", 32, 128, 512, 128, 3), ("layer2->3", 32, 128, 512, 256, 1),
                                   ("layer3 pair", 16, 256, 1024, 256, 22), ("layer3->4", 16, 256, 1024, 512, 1)):
    x, res, g1, g2 = t(k1, s), t(c1, s), t(c1, s), t(c2, s)
    wa = (torch.randn((c1, k1), device=dev) / k1 ** 0.5).to(bf); wb = (torch.randn((c2, c1), device=dev) / c1 ** 0.5).to(bf)
    wap, wbp = RN.frag_pack(wa), RN.frag_pack(wb)
    ba, bb = torch.randn(c1, device=dev), torch.randn(c2, device=dev)
    y1, y2 = empty_nhwc(B, c1, s, s, bf, dev), empty_nhwc(B, c2, s, s, bf, dev)
    f_chain = run(lambda: RN.conv1x1_chain(x, wap, ba, res, RELU, None, NONE, y1, wbp, bb, RELU, None, NONE, y2))
    f_two = run(lambda: (RN.conv1x1(x, wa, ba, y1, RELU, residual=res), RN.conv1x1(y1, wb, bb, y2, RELU)))
    b_chain = run(lambda: RN.conv1x1_chain(x, wap, None, res, NONE, g1, RELU, y1, wbp, None, NONE, g2, RELU, y2))
    b_two = run(lambda: (RN.conv1x1(x, wa, None, y1, NONE, residual=res, egate=g1, egate_act=RELU), RN.conv1x1(y1, wb, None, y2, NONE, egate=g2, egate_act=RELU)))
    m = B * s * s
    mb = (m * (k1 + 2 * c1 + c2) + k1 * c1 + c1 * c2) * 2 / 1e6
    for i, v in enumerate((f_chain, f_two, b_chain, b_two)): tot[i] += v * count
    print(f"{name:12s} M={m:6d} {k1:3d}->{c1:4d}->{c2:3d} x{count:2d}: forward chain {f_chain:6.1f} us / two launches {f_two:6.1f} us | backward chain {b_chain:6.1f} / {b_two:6.1f} us | {mb:6.1f} MB algorithmic = {mb / f_chain * 1e3:5.0f} GB/s (chain fwd)", flush=True)
print(f"whole network (30 pairs): forward chain {tot[0]:.0f} us vs {tot[1]:.0f} us; backward chain {tot[2]:.0f} us vs {tot[3]:.0f} us")
