"""Debug: which captured tensors of the (2,128,96) eval-mode step differ between option 3 = 0 and 2, and is the decoder's forward consistent with its captured input."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "weather-unet_amd")); sys.path.insert(0, root)
import numpy as np, torch, torch.nn.functional as F
from oracle import cunet_ref as O
from wu import unet_graph, _lib
import cunet
DEV = "cuda:0"
n, h, w = 2, 128, 96
nc, seed = 5, 14
p0 = O.make_cunet_params(nc, seed)
r = O._rng("stage", seed)
x = torch.from_numpy(r.uniform(-1, 1, size=(n, 3, h, w)).astype(np.float32))
c = torch.softmax(torch.from_numpy(r.standard_normal((n, nc)).astype(np.float32)), 1)
caps = {}
for opt in (0, 2):
    _lib.call("wu_set_option", 3, opt)
    net = cunet.Conditional_UNet(nc, precision="bf16"); net.load_state_dict(p0); net = net.to(DEV); net.train(False); net.dropout_seed = 33
    cap = {}; unet_graph.CAPTURE = cap
    xd = x.to(DEV); out = net(xd, c.to(DEV)); torch.mean(torch.abs(out - xd)).backward(); torch.cuda.synchronize()
    unet_graph.CAPTURE = None
    T = {k: v.detach().float().cpu() for k, v in cap.items() if torch.is_tensor(v)}
    T.update({"grad:" + k: q.grad.detach().float().cpu() for k, q in net.named_parameters() if q.grad is not None})
    caps[opt] = T
    # forward consistency of the up3 block: relu(conv2(bf16(relu(conv1(cat3))))) against the captured u3b
    wq = lambda k: p0[k].to(torch.bfloat16).float()
    mid = torch.relu(F.conv2d(T["cat3"], wq("dconv_up3.0.weight"), p0["dconv_up3.0.bias"], padding=1)).to(torch.bfloat16).float()
    y = torch.relu(F.conv2d(mid, wq("dconv_up3.2.weight"), p0["dconv_up3.2.bias"], padding=1))
    d = (y - T["u3b"]).abs()
    print(f"opt3={opt}: u3b vs fp32 recomputation from captured cat3: max {d.max():.4f} mean {d.mean():.6f} scale {y.abs().max():.3f}; rows with err > 0.05: {(d > 0.05).sum().item()}")
    bad = (d > 0.05).nonzero()
    if len(bad):
        print("   first bad (n, c, y, x):", bad[:8].tolist(), " distinct y:", sorted(set(bad[:, 2].tolist()))[:40], " distinct x:", sorted(set(bad[:, 3].tolist()))[:40])
for k in caps[0]:
    a, b = caps[0][k], caps[2][k]
    rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
    print(f"{k:32s} rel diff {rel:.5f}")

print("---- stage dconv_up3: HIP gradients against an fp32 recomputation from the captured stage input / upstream gradient ----")
def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
for opt in (0, 2):
    T = caps[opt]
    w0 = p0["dconv_up3.0.weight"].to(torch.bfloat16).float().requires_grad_(True); b0 = p0["dconv_up3.0.bias"].clone().requires_grad_(True)
    w2 = p0["dconv_up3.2.weight"].to(torch.bfloat16).float().requires_grad_(True); b2 = p0["dconv_up3.2.bias"].clone().requires_grad_(True)
    xin = T["cat3"].clone().requires_grad_(True)
    mid = torch.relu(F.conv2d(xin, w0, b0, padding=1))
    y = F.conv2d(mid.to(torch.bfloat16).float() + (mid - mid.detach()), w2, b2, padding=1)        # no final ReLU: g_u3b is pre-gated by HIP's own u3b > 0
    gs = torch.autograd.grad(y, [xin, w0, b0, w2, b2], T["g_u3b"])
    names = ["d input", "0.weight", "0.bias", "2.weight", "2.bias"]
    hip = [T["g_cat3"], T["grad:dconv_up3.0.weight"], T["grad:dconv_up3.0.bias"], T["grad:dconv_up3.2.weight"], T["grad:dconv_up3.2.bias"]]
    print(f"opt3={opt}: " + "  ".join(f"{nm} {cos(a, b):.6f}" for nm, a, b in zip(names, hip, gs)))
    print(f"         sum(g_u3b) per channel vs HIP db2: rel {((T['g_u3b'].sum((0, 2, 3)) - hip[4]).norm() / hip[4].norm()).item():.5f};  g_u3b nonzero where u3b == 0: {((T['g_u3b'] != 0) & (T['u3b'] <= 0)).sum().item()}")
    # the oracle's emulation of the block, as the test runs it
    pp = {k: p0[k].clone().requires_grad_(True) for k in ("dconv_up3.0.weight", "dconv_up3.0.bias", "dconv_up3.2.weight", "dconv_up3.2.bias")}
    xin2 = T["cat3"].clone().requires_grad_(True)
    ye = O.r_double_conv(pp, "dconv_up3", xin2, emu=True)
    ge = torch.autograd.grad(ye, [xin2] + list(pp.values()), T["g_u3b"])
    print(f"         oracle emulation: " + "  ".join(f"{nm} {cos(a, b):.6f}" for nm, a, b in zip(names, hip, ge)))
    dis = ((ye > 0) != (T["u3b"] > 0))
    print(f"         gate disagreements emulated y > 0 vs HIP u3b > 0: {dis.sum().item()} of {dis.numel()}; |g_u3b| mass on them: {(T['g_u3b'].abs() * dis).sum().item() / T['g_u3b'].abs().sum().item():.5f}; max |y_emu - u3b| {(ye - T['u3b']).abs().max().item():.4f}")
