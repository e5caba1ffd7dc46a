import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch, cunet
from wu import unet_graph as UG
dev = torch.device("cuda")
def mk():
    torch.manual_seed(3)
    net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()
    net.dropout_seed = 5
    return net
g = torch.Generator().manual_seed(11)
x = (torch.rand((32, 3, 256, 256), generator=g) * 2 - 1).to(dev)
c = torch.eye(5)[torch.arange(32) % 5].to(dev)
def run(net, bwd):
    for p in net.parameters(): p.grad = None
    out = net(x, c)
    if bwd: torch.mean(torch.abs(out - x)).backward()
    torch.cuda.synchronize()
    return out.detach().clone()
for bits in (True, False):
    for bwd in (False, True):
        UG.GATE_BITS = bits
        net = mk()
        o = [run(net, bwd) for _ in range(4)]
        d = [(o[0] - oi).abs().max().item() for oi in o[1:]]
        nz = [(o[0] != oi).sum().item() for oi in o[1:]]
        print(f"bits={bits} bwd={bwd}: max diffs vs run0 {d}, #different {nz}")
