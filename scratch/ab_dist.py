

def main():
    import sys, os, time, statistics
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, cunet
    import torch.distributed as dist
    from wu.ddp import GradBucketReducer, ready_order
    dev = torch.device('cuda:0'); torch.cuda.set_device(0)
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
    x = (torch.rand((32, 3, 256, 256)) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(32) % 5].to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.0, 0.999), weight_decay=5e-6, fused=True)
    def step_plain():
        opt.zero_grad(set_to_none=True)
        torch.mean(torch.abs(net(x, c) - x)).backward()
        opt.step()
    def bench(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"plain, no process group: {bench(step_plain):.3f} ms")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print(f"plain, nccl group initialised: {bench(step_plain):.3f} ms")
    red = GradBucketReducer(ready_order(net), bucket_mb=12.0, ready_order=True).attach(net)
    def step_ddp():
        red.zero_grad()
        torch.mean(torch.abs(net(x, c) - x)).backward()
        red.finalize()
        opt.step()
    print(f"reducer + sink, nccl group initialised: {bench(step_ddp):.3f} ms")
    net.grad_sink = None
    print(f"reducer via autograd hooks only: {bench(step_ddp):.3f} ms")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
