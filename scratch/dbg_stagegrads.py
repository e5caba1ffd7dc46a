

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, torch.nn.functional as F
    from oracle import cunet_ref as O
    import cunet
    DEV='cuda:0'
    nc, seed = 5, 1
    x, c = O.make_inputs(2, 64, nc, seed, True)
    def oracle(dt):
        p={k:v.clone().to(dt) for k,v in O.make_cunet_params(nc,seed).items()}
        grads={}
        def keep(name,t):
            t.retain_grad(); grads[name]=t; return t
        xx=x.to(dt).requires_grad_(True); cc=c.to(dt)
        def dc(name,t):
            a=keep(name+'.0',F.relu(F.conv2d(t,p[name+'.0.weight'],p[name+'.0.bias'],padding=1)))
            return keep(name+'.2',F.relu(F.conv2d(a,p[name+'.2.weight'],p[name+'.2.bias'],padding=1)))
        conv1=dc('dconv_down1',xx); t=F.max_pool2d(conv1,2)
        conv2=dc('dconv_down2',t); t=F.max_pool2d(conv2,2)
        conv3=dc('dconv_down3',t); t=F.max_pool2d(conv3,2)
        t=dc('dconv_down4',t)
        t=keep('cat3',torch.cat([O.upsample2(O.adain(p,'adain3',t,cc)),conv3],1)); t=dc('dconv_up3',t)
        t=keep('cat2',torch.cat([O.upsample2(O.adain(p,'adain2',t,cc)),conv2],1)); t=dc('dconv_up2',t)
        t=keep('cat1',torch.cat([O.upsample2(O.adain(p,'adain1',t,cc)),conv1],1)); t=dc('dconv_up1',t)
        out=torch.tanh(F.conv2d(t,p['conv_last.weight'],p['conv_last.bias']))
        O.bench_loss(out,xx).backward()
        return {k:v.grad.double() for k,v in grads.items()}
    g32=oracle(torch.float32); g64=oracle(torch.float64)
    net = cunet.Conditional_UNet(nc, precision='fp32'); net.load_state_dict(O.make_cunet_params(nc, seed)); net=net.to(DEV).eval()
    got={}
    def hook_out(name):
        def fh(mod, inp, out):
            out.register_hook(lambda g, name=name: got.__setitem__(name, g.detach().float().cpu().double()))
        return fh
    for blk in ['dconv_down1','dconv_down2','dconv_down3','dconv_down4','dconv_up3','dconv_up2','dconv_up1']:
        getattr(net,blk)[0].register_forward_hook(hook_out(blk+'.0'))
        getattr(net,blk).register_forward_hook(hook_out(blk+'.2'))
    xd = x.to(DEV); out = net(xd, c.to(DEV)); torch.mean(torch.abs(out - xd)).backward()
    def rel(a,b): return ((a.reshape(-1)-b.reshape(-1)).norm()/b.norm()).item()
    for k in ['dconv_up1.2','dconv_up1.0','dconv_up2.2','dconv_up2.0','dconv_up3.2','dconv_up3.0','dconv_down4.2','dconv_down4.0','dconv_down3.2','dconv_down3.0','dconv_down2.2','dconv_down2.0','dconv_down1.2','dconv_down1.0']:
        print(f'd/d out[{k:14s}] gpu-vs-f64 {rel(got[k],g64[k]):.3e}  cpu32-vs-f64 {rel(g32[k],g64[k]):.3e}  shape {tuple(got[k].shape)}')


if __name__ == "__main__":
    main()
