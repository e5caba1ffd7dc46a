

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, numpy as np
    from oracle import cunet_ref as O
    import cunet
    DEV='cuda:0'
    nc, seed = 5, 1
    x, c = O.make_inputs(2, 64, nc, seed, True)
    res={}
    for dt in [torch.float32, torch.float64]:
        p={k:v.clone().to(dt).requires_grad_(True) for k,v in O.make_cunet_params(nc,seed).items()}
        out=O.cunet_forward(p,x.to(dt),c.to(dt)); O.bench_loss(out,x.to(dt)).backward()
        res[dt]={k:v.grad.double() for k,v in p.items() if v.grad is not None}
    net = cunet.Conditional_UNet(nc, precision='fp32'); net.load_state_dict(O.make_cunet_params(nc, seed)); net=net.to(DEV).eval()
    xd = x.to(DEV); out = net(xd, c.to(DEV)); torch.mean(torch.abs(out - xd)).backward()
    def rel(a,b): return ((a.reshape(-1)-b.reshape(-1)).norm()/b.norm()).item()
    for k, prm in net.named_parameters():
        if prm.grad is None or 'bias' in k: continue
        g=prm.grad.detach().cpu().double()
        print(f'{k:26s} gpu-vs-f64 {rel(g,res[torch.float64][k]):.3e}   cpu32-vs-f64 {rel(res[torch.float32][k],res[torch.float64][k]):.3e}   gpu-vs-cpu32 {rel(g,res[torch.float32][k]):.3e}')


if __name__ == "__main__":
    main()
