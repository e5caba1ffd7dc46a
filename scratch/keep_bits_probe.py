#!/usr/bin/env python3
"""Dropout keep decisions: stored bytes (default) vs drawn again by the backward (WU_KEEP_BITS=0).  Correctness of the switch at the full step shape
(B=32 256x256 bf16): same output, gradients compared tensor by tensor.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "weather-unet_amd")]
import torch
import cunet
from wu import unet_graph as UG

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()
g = torch.Generator().manual_seed(1)
B = int(os.environ.get("PROBE_B", "32"))
x = (torch.rand((B, 3, 256, 256), generator=g) * 2 - 1).to(dev)
c = torch.eye(5)[torch.arange(B) % 5].to(dev)
res = []
for flag in (True, False, True):
    UG.KEEP_BITS_STORED = flag
    net.dropout_seed = 7
    net.zero_grad(set_to_none=True)
    out = net(x, c)
    torch.mean(torch.abs(out - x)).backward()
    torch.cuda.synchronize()
    res.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
print("output stored vs re-drawn equal:", torch.equal(res[0][0], res[1][0]), "| stored twice equal:", torch.equal(res[0][0], res[2][0]))
worst = 0.0
nbit = 0
for k in res[0][1]:
    a, b, a2 = res[0][1][k].double(), res[1][1][k].double(), res[2][1][k].double()
    assert torch.equal(a, a2), f"stored path not reproducible: {k}"
    rel = ((a - b).norm() / (a.norm() + 1e-300)).item()
    worst = max(worst, rel)
    nbit += int(torch.equal(a, b))
    if rel > 0:
        print(f"  {k:32s} rel-L2 diff {rel:.3e}  max-abs {float((a - b).abs().max()):.3e}")
print(f"gradients: {nbit} of {len(res[0][1])} bit-identical, worst rel-L2 diff {worst:.3e}")
