"""Worker of tests/test_gpu_round3.py::test_two_process_data_parallel_real_kernels -- NOT collected by pytest.

Started as ``python -m torch.distributed.run --nproc-per-node 2 tests/ddp_two_rank_worker.py``: two ranks, both on cuda:0, backend
gloo (RCCL refuses two ranks on one device; gloo all-reduces CUDA tensors through host memory).  Everything else is the product
path of an N-GPU run: the real HIP kernels, the fused U-Net node's gradient sink writing into the reducer's bucket
views, the side stream for the weight gradients, the bucket collectives launched from inside backward, the GAN iteration's two
reducers (wu/train_step.py), the SN-buffer broadcast.

Rank 0 also runs the SAME work in one process on the concatenated batch and writes the comparison to $WU_DDP_OUT (JSON).
Dropout masks are injected (``net.dropout_masks``) so that a rank's shard sees exactly the rows of the global masks."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "weather-unet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import cunet_ref as O  # noqa: E402  (deterministic parameter / input fill only: no oracle compute is used here)

DEV = "cuda:0"
NC = 5


def masks_for(batch, size, seed):
    g = torch.Generator().manual_seed(seed)
    s = size // 4
    return tuple((torch.rand((batch, ch, hw, hw), generator=g) < 0.7).to(torch.uint8)
                 for ch, hw in ((512, s), (256, 2 * s), (128, 4 * s)))


def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def gathered_equal(t, world):
    """True if the flat CPU tensor `t` is bitwise identical on every rank."""
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return all(torch.equal(parts[0], q) for q in parts[1:])


def flat(tensors):
    return torch.cat([t.detach().float().reshape(-1).cpu() for t in tensors])


def unet_part(rank, world, precision, res):
    import cunet
    from wu.ddp import GradBucketReducer, ready_order, shard_batch
    batch, size, seed = 4, 64, 6
    x, c = O.make_inputs(batch, size, NC, seed, True)
    masks = masks_for(batch, size, 31)
    net = cunet.Conditional_UNet(NC, precision=precision)
    net.load_state_dict(O.make_cunet_params(NC, seed))
    net = net.to(DEV).train()
    red = GradBucketReducer(ready_order(net), bucket_mb=2.0, ready_order=True).attach(net)
    xs, cs = shard_batch(x, rank, world).to(DEV), shard_batch(c, rank, world).to(DEV)
    net.dropout_masks = tuple(shard_batch(m, rank, world) for m in masks)
    red.zero_grad()
    out = net(xs, cs)
    torch.mean(torch.abs(out - xs)).backward()
    launched_in_backward = len(red.launch_log)
    red.finalize()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad and not k.endswith("emb.weight")}
    agree = gathered_equal(flat(grads.values()), world)
    finite = all(torch.isfinite(g).all().item() for g in grads.values())
    if precision == "bf16":
        res["bf16_step_finite"] = finite
        res["bf16_ranks_agree_bitwise"] = agree
        return
    res["ranks_agree_bitwise"] = agree
    res["unet_buckets_launched_during_backward"] = launched_in_backward
    res["unet_buckets"] = len(red.buckets)
    if rank == 0:
        single = cunet.Conditional_UNet(NC, precision=precision)
        single.load_state_dict(O.make_cunet_params(NC, seed))
        single = single.to(DEV).train()
        single.dropout_masks = masks
        xd = x.to(DEV)
        torch.mean(torch.abs(single(xd, c.to(DEV)) - xd)).backward()
        worst, who = 0.0, None
        for k, p in single.named_parameters():
            if p.grad is None:
                continue
            r = rel(grads[k], p.grad)
            if r > worst:
                worst, who = r, k
        res["unet_worst_rel"], res["unet_worst_key"] = worst, who


def gan_part(rank, world, res):
    from wu.ddp import shard_batch
    from wu.train_step import WeatherTransferStep
    batch, size, seed = 4, 64, 9
    x, _ = O.make_inputs(batch, size, NC, seed, True)
    xr, _ = O.make_inputs(batch, size, NC, seed + 1, True)
    masks = masks_for(batch, size, 32)

    def make(ddp):
        st = WeatherTransferStep(NC, mode="cls", precision="fp32", device=DEV, ddp=ddp, seed=1)
        st.inference.load_state_dict(O.make_cunet_params(NC, seed))
        st.discriminator.load_state_dict(O.make_sndisc_params(NC, seed))
        return st

    st = make(True)
    assert st.g_red is not None and st.d_red is not None
    st.inference.dropout_masks = tuple(shard_batch(m, rank, world) for m in masks)
    xs, xrs = shard_batch(x, rank, world).to(DEV), shard_batch(xr, rank, world).to(DEV)
    losses = st.step(xs, xrs)
    torch.cuda.synchronize()
    g_grads = {k: p.grad.detach().clone() for k, p in st.inference.named_parameters() if not k.endswith("emb.weight")}
    d_grads = {k: p.grad.detach().clone() for k, p in st.discriminator.named_parameters()}
    st.step(xs, xrs)                                     # second iteration, weights now moved by two Adam steps each
    torch.cuda.synchronize()
    params = flat(list(st.inference.parameters()) + list(st.discriminator.parameters()))
    bufs = flat([v for k, v in st.discriminator.state_dict().items() if k.endswith(("weight_u", "weight_v"))])
    res["params_agree_after_steps"] = gathered_equal(params, world)
    res["sn_buffers_agree_bitwise"] = gathered_equal(bufs, world)
    res["gan_losses_rank0"] = [float(v) for v in losses]
    if rank == 0:
        one = make(False)
        one.inference.dropout_masks = masks
        one.step(x.to(DEV), xr.to(DEV))
        worst_g = max(rel(g_grads[k], p.grad) for k, p in one.inference.named_parameters() if p.grad is not None)
        worst_d = max(rel(d_grads[k], p.grad) for k, p in one.discriminator.named_parameters() if p.grad is not None)
        res["gan_g_worst_rel"], res["gan_d_worst_rel"] = worst_g, worst_d


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"world": world}
    try:
        unet_part(rank, world, "fp32", res)
        unet_part(rank, world, "bf16", res)
        gan_part(rank, world, res)
        dist.barrier()
        if rank == 0:
            with open(os.environ["WU_DDP_OUT"], "w") as fh:
                json.dump(res, fh)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
