

def main():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, cunet
    from torch.profiler import profile, ProfilerActivity
    dev = torch.device('cuda:0')
    net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
    x = (torch.rand((32, 3, 256, 256)) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(32) % 5].to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.0, 0.999), fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        torch.mean(torch.abs(net(x, c) - x)).backward()
        opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.count)
    for e in rows[:40]:
        print(f"{e.key[:60]:60s} count {e.count:4d}  cuda {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)):8.0f} us")


if __name__ == "__main__":
    main()
