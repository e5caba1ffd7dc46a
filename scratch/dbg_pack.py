import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch, cunet
dev = torch.device("cuda")
torch.manual_seed(0)
net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()
x = (torch.rand(4, 3, 64, 64, device=dev) * 2 - 1); c = torch.eye(5, device=dev)[torch.arange(4) % 5]
for fused in (True, False):
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.0, 0.999), fused=fused)
    w = net.dconv_down2[0].weight
    pc = net.dconv_down2[0]._packed
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        out = net(x, c)
        loss = out.abs().mean(); loss.backward()
        v0, k0 = w._version, pc.key
        wf_before = pc.w_fwd.clone()
        opt.step()
        net(x, c)   # next forward repacks if the key changed
        same_pack = torch.equal(wf_before, pc.w_fwd)
        # what the packed weights SHOULD be
        ref = w.detach().permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).to(torch.bfloat16)
        ok = torch.equal(ref, pc.w_fwd)
        print(f"fused={fused} it={it}: version {v0} -> {w._version}, key changed {k0 != pc.key}, packed unchanged {same_pack}, packed == current weight {ok}")
