"""Longer-run sanity of the GAN iteration (t_cls_train step: D update + G update, fused Adam, bf16, dropout on, stand-in estimator):
150 iterations on a fixed synthetic batch must stay finite, and two identical runs must end with bit-identical G and D parameters."""


def main():
    import sys, os, hashlib, math
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch
    from wu.train_step import WeatherTransferStep
    dev = torch.device('cuda:0')

    def run():
        st = WeatherTransferStep(5, mode="cls", precision="bf16", lr=1e-4, device=dev, ddp=False, seed=3)
        st.inference.dropout_seed = None
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(1)
        x = (torch.rand((8, 3, 128, 128), generator=g) * 2 - 1).to(dev)
        xr = (torch.rand((8, 3, 128, 128), generator=g) * 2 - 1).to(dev)
        log = []
        for it in range(150):
            out = st.step(x, xr)
            if it % 25 == 0 or it == 149:
                vals = [float(v) for v in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(v) and v.numel() == 1]
                log.append((it, [round(v, 5) for v in vals]))
                assert all(math.isfinite(v) for v in vals), (it, vals)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for net in (st.inference, st.discriminator):
            for k, v in sorted(net.state_dict().items()):
                assert torch.isfinite(v.float()).all(), k
                h.update(v.detach().float().cpu().numpy().tobytes())
        return log, h.hexdigest()[:16]

    import torch
    # identical seeds for the dropout counter too: reset the module-level counter between runs
    import cunet
    import itertools
    cunet._SEED_COUNTER = itertools.count(1)
    l1, h1 = run()
    cunet._SEED_COUNTER = itertools.count(1)
    l2, h2 = run()
    print("losses", l1)
    print("param hash run 1", h1, "run 2", h2, "identical:", h1 == h2)
    print("soak_gan OK" if all(len(v) > 0 for _, v in l1) else "soak_gan: no scalar outputs logged")


if __name__ == "__main__":
    main()
