"""Longer-run sanity: 300 Adam steps on a fixed synthetic batch (bf16, two-stream backward, dropout on): the loss must stay
finite and fall; two identical runs must produce bit-identical parameters (determinism end to end)."""


def main():
    import sys, os, hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'weather-unet_amd'))
    import torch, cunet
    dev = torch.device('cuda:0')
    def run():
        torch.manual_seed(0)
        net = cunet.Conditional_UNet(5, precision='bf16').to(dev).train()
        net.dropout_seed = 1234
        g = torch.Generator().manual_seed(1)
        x = (torch.rand((8, 3, 128, 128), generator=g) * 2 - 1).to(dev); c = torch.eye(5)[torch.arange(8) % 5].to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.0, 0.999), fused=True)
        losses = []
        for it in range(300):
            opt.zero_grad(set_to_none=True)
            loss = torch.mean(torch.abs(net(x, c) - x))
            loss.backward()
            opt.step()
            if it % 50 == 0 or it == 299: losses.append(round(loss.item(), 5))
        h = hashlib.sha256()
        for p in net.parameters(): h.update(p.detach().cpu().numpy().tobytes())
        return losses, h.hexdigest()[:16]
    a = run(); b = run()
    print("losses", a[0]); print("param hash run 1", a[1], "run 2", b[1], "identical:", a[1] == b[1])
    assert all(l == l and l < 10 for l in a[0]) and a[0][-1] < a[0][0] and a[1] == b[1]
    print("soak OK")


if __name__ == "__main__":
    main()
