

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, torch.nn.functional as F
    from wu import functional as WF
    from wu.layout import as_nhwc, precision_code
    dev=torch.device('cuda:0')
    def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm()).item()
    for (n,cin,cout,h,w) in [(2,64,64,64,64),(2,192,64,64,64),(2,64,64,8,8),(1,128,128,16,16)]:
      for act in [0,1]:
        g=torch.Generator().manual_seed(0)
        x=(torch.rand((n,cin,h,w),generator=g)*2-1).double().requires_grad_(True)
        wt=((torch.rand((cout,cin,3,3),generator=g)*2-1)*0.1).double().requires_grad_(True)
        b=(torch.rand((cout,),generator=g)*2-1).double().requires_grad_(True)
        pre=F.conv2d(x,wt,b,padding=1); ref=F.relu(pre) if act else pre
        gy=(torch.rand(ref.shape,generator=g)*2-1).double()
        ref.backward(gy)
        xg=as_nhwc(x.detach().float().to(dev),0).requires_grad_(True); wg=wt.detach().float().to(dev).requires_grad_(True); bg=b.detach().float().to(dev).requires_grad_(True)
        y=WF.conv3x3(xg,wg,bg,WF.PackedConv(),1,act)
        y.backward(as_nhwc(gy.float().to(dev),0))
        # fraction of relu disagreements
        dis=((y.detach().cpu()>0)!=(ref.detach()>0)).sum().item() if act else 0
        print((n,cin,cout,h,w),'act',act,'fwd',rel(y.detach().cpu(),ref.detach()),'dx',rel(xg.grad.cpu(),x.grad),'dw',rel(wg.grad.cpu(),wt.grad),'db',rel(bg.grad.cpu(),b.grad),'relu flips',dis)


if __name__ == "__main__":
    main()
