

def main():
    import sys, os, time
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, cunet
    import torch.distributed as dist
    from wu import unet_graph as UG
    from wu.ddp import GradBucketReducer, ready_order
    dev = torch.device('cuda:0'); torch.cuda.set_device(0)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29535"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
    x = (torch.rand((32, 3, 256, 256)) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(32) % 5].to(dev)
    red = GradBucketReducer(ready_order(net), bucket_mb=12.0, ready_order=True).attach(net)
    opt = torch.optim.Adam(list(net.parameters()), lr=1e-4, betas=(0.0, 0.999), weight_decay=5e-6, fused=True)
    def step():
        red.zero_grad()
        torch.mean(torch.abs(net(x, c) - x)).backward()
        red.finalize()
        opt.step()
    def bench(n=10):
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f'auto-picked side stream: {bench():.3f} ms/step (handle {UG._side_stream(dev).cuda_stream:#x})')
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    for k, s in enumerate(streams):
        UG._SIDE[(dev, torch.cuda.current_stream(dev).cuda_stream)] = s
        print(f"side stream candidate {k} (handle {s.cuda_stream:#x}): {bench():.3f} ms/step")
    UG.SIDE_STREAM_WGRAD = False
    print(f"single stream: {bench():.3f} ms/step")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
