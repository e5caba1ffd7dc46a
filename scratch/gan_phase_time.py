"""Where does a GAN iteration's wall time go -- GPU work or host launch overhead?  Per phase of WeatherTransferStep.step() (ResNet-101
estimator, B=32, 256x256, bf16): `enqueue` = host time until Python returns (nothing waited for), `wall` = time until the GPU has
finished the phase (synchronised before and after).  A phase with enqueue ~ wall is host-bound: faster kernels cannot shorten it.

    python scratch/gan_phase_time.py [cls|est] [batch]
"""
import os
import statistics
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

import ops  # noqa: E402
from wu.resnet import ResNet101Estimator  # noqa: E402
from wu.train_step import WeatherTransferStep  # noqa: E402
from wu.unet_graph import prepare_side_stream  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "cls"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    est = ResNet101Estimator(5, precision="bf16")
    T = WeatherTransferStep(5, mode=mode, precision="bf16", device=dev, ddp=False, seed=0, estimator=est)
    g = torch.Generator(device="cpu").manual_seed(1000)
    x = (torch.rand((B, 3, 256, 256), generator=g) * 2 - 1).to(dev)
    xr = (torch.rand((B, 3, 256, 256), generator=g) * 2 - 1).to(dev)
    prepare_side_stream(dev)
    for _ in range(3):
        T.step(x, xr)
    torch.cuda.synchronize()

    rec = {}

    def phase(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rec.setdefault(name, []).append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
        return out

    D, G = T.discriminator, T.inference

    def one():
        def est_pass():
            with torch.no_grad():
                raw_all = T.estimator_(torch.cat([xr, x]))
                raw, raw_img = raw_all[:B], raw_all[B:]
                sm = (lambda t: torch.softmax(t, dim=1)) if mode == "cls" else (lambda t: t)
                return sm(raw), sm(raw_img)
        rand_labels, pred_labels = phase("1 estimator no-grad (2B)", est_pass)
        T.d_opt.zero_grad(set_to_none=True)

        def g_nograd():
            with torch.no_grad():
                return G(x, rand_labels)
        fake = phase("2 G fwd no-grad", g_nograd)
        real_d = phase("3 D fwd real", lambda: D(x, pred_labels)[0])
        fake_d = phase("4 D fwd fake", lambda: D(fake, rand_labels)[0])
        d_loss = phase("5 d loss", lambda: ops.dis_hinge(fake_d, real_d))
        phase("6 d backward", lambda: d_loss.backward())
        phase("7 d_opt.step", lambda: T.d_opt.step())
        T.g_opt.zero_grad(set_to_none=True)
        dp = list(D.parameters())
        for p in dp:
            p.requires_grad_(False)
        fake2 = phase("8 G fwd", lambda: G(x, rand_labels))
        fd2 = phase("9 D fwd (G update)", lambda: D(fake2, rand_labels)[0])
        for p in dp:
            p.requires_grad_(True)
        fc = phase("10 estimator fwd (grad)", lambda: T.estimator(fake2))

        def losses():
            adv = ops.gen_hinge(fd2)
            w = ops.pred_loss(fc, rand_labels, one_hot=False)
            diff = torch.mean(torch.abs(fake2 - x), [1, 2, 3])
            lmda = torch.mean(torch.abs(pred_labels - rand_labels), 1)
            return adv + torch.mean(diff / (lmda + 1e-7)) + w
        g_loss = phase("11 g losses", losses)
        phase("12 g backward", lambda: g_loss.backward())
        phase("13 g_opt.step", lambda: T.g_opt.step())

    for _ in range(7):
        one()
    # the un-phased step: host enqueue time of a whole iteration against its wall time
    whole = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.step(x, xr)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        whole.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(f"{'phase':28s} {'enqueue ms':>11s} {'wall ms':>9s}")
    se = sw = 0.0
    for k, v in rec.items():
        e, w = statistics.median(a for a, _ in v), statistics.median(b for _, b in v)
        se += e
        sw += w
        print(f"{k:28s} {e:11.3f} {w:9.3f}" + ("   <- host-bound" if e > 0.8 * w else ""))
    print(f"{'sum of phases':28s} {se:11.3f} {sw:9.3f}")
    print(f"{'whole step() (no phase syncs)':28s} {statistics.median(a for a, _ in whole):11.3f} {statistics.median(b for _, b in whole):9.3f}")


if __name__ == "__main__":
    main()
