

def main():
    import sys, os, time, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, cunet
    from wu import unet_graph as UG
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
    x = (torch.rand((32, 3, 256, 256)) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(32) % 5].to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.0, 0.999), fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        torch.mean(torch.abs(net(x, c) - x)).backward()
        opt.step()
    res = {0: [], 1: []}
    for rnd in range(6):
        for v in (0, 1):
            UG.SIDE_STREAM_WGRAD = bool(v)
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"single stream {statistics.median(res[0]):.3f} ms/step   wgrad on side stream {statistics.median(res[1]):.3f} ms/step")
    # --- with the gradient sink (single process: collectives are no-ops)
    from wu.ddp import GradBucketReducer, ready_order
    red = GradBucketReducer(ready_order(net), bucket_mb=12.0, ready_order=True).attach(net)
    def step_ddp():
        red.zero_grad()
        torch.mean(torch.abs(net(x, c) - x)).backward()
        red.finalize()
        opt.step()
    res = {0: [], 1: []}
    for rnd in range(6):
        for v in (0, 1):
            UG.SIDE_STREAM_WGRAD = bool(v)
            for _ in range(3): step_ddp()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): step_ddp()
            torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"sink: single stream {statistics.median(res[0]):.3f} ms/step   wgrad on side stream {statistics.median(res[1]):.3f} ms/step")


if __name__ == "__main__":
    main()
