import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import as_nhwc, empty_nhwc
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, cin, cout, h, w = 2, 64, 128, 16, 32
bf = torch.bfloat16
wt = (torch.rand((cout, cin, 3, 3)) * 0.2 - 0.1).to(bf).float()
gy = (torch.rand((n, cout, h // 2, w // 2)) * 2 - 1).to(bf).float()
xg = (torch.rand((n, cin, h, w)) * 2 - 1).to(bf).float()
_, wd = K.pack_conv3x3(wt.to(dev), _lib.BF16)
gyd = as_nhwc(gy.to(dev), _lib.BF16)
eg = as_nhwc(xg.to(dev), _lib.BF16)
out = {}
for name, o15, gated in (("tap", 0, True), ("gath", 10, True), ("gath_nogate", 10, False), ("tap_nogate", 0, False)):
    _lib.call("wu_set_option", 15, o15)
    dx = empty_nhwc(n, cin, h, w, bf, dev); dx.fill_(float("nan"))
    K.conv3x3_s2_dgrad(gyd, wd, dx, egate=eg if gated else None, egate_act=K.ACT_LEAKY if gated else K.ACT_NONE)
    torch.cuda.synchronize()
    out[name] = dx.float().cpu()
print("nogate diff", (out["gath_nogate"] - out["tap_nogate"]).abs().max().item())
d = (out["gath"] - out["tap"]).abs()
print("gate diff", d.max().item(), "count", int((d > 0.05).sum()), "of", d.numel())
bad = (d > 0.05).nonzero()
print(bad[:20])
import collections
print("by (y%2,x%2):", collections.Counter((int(b[2]) % 2, int(b[3]) % 2) for b in bad))
print("by channel//8:", sorted(collections.Counter(int(b[1]) // 8 for b in bad).items()))
print("by n:", collections.Counter(int(b[0]) for b in bad))
# is gath == nogate * gate-from-somewhere?
ratio = out["gath"] / out["gath_nogate"]
print("ratio values sample", ratio[bad[0][0], bad[0][1], bad[0][2], bad[0][3]].item() if len(bad) else None)
