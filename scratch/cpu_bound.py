

def main():
    import sys, os, time
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, cunet
    import torch.distributed as dist
    from wu.ddp import GradBucketReducer, ready_order
    dev = torch.device('cuda:0'); torch.cuda.set_device(0)
    use_dist = len(sys.argv) > 1 and sys.argv[1] == "dist"
    if use_dist:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29534"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
    x = (torch.rand((32, 3, 256, 256)) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(32) % 5].to(dev)
    params = list(net.parameters())
    red = GradBucketReducer(ready_order(net), bucket_mb=12.0, ready_order=True).attach(net) if use_dist else None
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.0, 0.999), weight_decay=5e-6, fused=True)
    def step():
        if red is not None: red.zero_grad()
        else: opt.zero_grad(set_to_none=True)
        torch.mean(torch.abs(net(x, c) - x)).backward()
        if red is not None: red.finalize()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{'ddp+sink (nccl init)' if use_dist else 'plain'}: CPU enqueue {(t1 - t0) / 20 * 1e3:.2f} ms/step, total {(t2 - t0) / 20 * 1e3:.2f} ms/step")
    if use_dist: dist.destroy_process_group()


if __name__ == "__main__":
    main()
