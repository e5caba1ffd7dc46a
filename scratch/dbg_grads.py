

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, numpy as np
    from oracle import cunet_ref as O
    import cunet
    DEV='cuda:0'
    nc, seed = 5, 1
    for precision in ['fp32','bf16']:
        net = cunet.Conditional_UNet(nc, precision=precision); net.load_state_dict(O.make_cunet_params(nc, seed)); net=net.to(DEV).eval()
        p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
        x, c = O.make_inputs(2, 64, nc, seed, True)
        ref = O.cunet_forward(p, x, c); O.bench_loss(ref, x).backward()
        xd = x.to(DEV); out = net(xd, c.to(DEV)); torch.mean(torch.abs(out - xd)).backward()
        print(precision, 'fwd err', (out.detach().cpu()-ref.detach()).abs().max().item())
        # sign flips in loss: d|out-x|/dout = sign(out-x): count sign disagreements
        sg = torch.sign(out.detach().cpu()-x); sr = torch.sign(ref.detach()-x)
        print('  sign flips in |out-x|:', (sg!=sr).sum().item(), 'of', sg.numel())
        for k, prm in net.named_parameters():
            if prm.grad is None: continue
            a, b = prm.grad.detach().cpu().reshape(-1).double(), p[k].grad.reshape(-1).double()
            print(f'  {k:28s} rel {((a-b).norm()/b.norm()).item():.3e} cos {(torch.dot(a,b)/(a.norm()*b.norm())).item():.6f} |ref| {b.norm().item():.3e}')


if __name__ == "__main__":
    main()
