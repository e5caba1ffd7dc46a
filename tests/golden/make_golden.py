#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 -O tests/golden/make_golden.py

``-O`` strips the reference's ``assert isinstance(x, torch.cuda.FloatTensor)`` (utils.py:35) so
AdaIN runs on CPU; ``torchvision`` (imported by utils.py:5 only for the off-path ``Denormalize``)
is stubbed in ``sys.modules``.  The reference files are imported, never copied.

For every case the script (1) loads the build-owned deterministic parameters into the reference
module, (2) runs the reference, (3) runs the oracle restatement (oracle/cunet_ref.py) on the
same tensors and REQUIRES max-abs 0.0 between the two, (4) stores inputs-by-seed + reference
outputs (full tensors where small, else mean / abs-max / strided samples) as .npz fixtures.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
import cunet as ref_cunet  # noqa: E402  (the reference)
import disc as ref_disc    # noqa: E402
import ops as ref_ops      # noqa: E402

from oracle import cunet_ref as O  # noqa: E402

torch.set_num_threads(8)
NC = 5


def summary(t, nsamp=64):
    t = t.detach().reshape(-1).double()
    idx = torch.linspace(0, t.numel() - 1, nsamp).long()
    return np.concatenate([[t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()],
                           t[idx].numpy()]).astype(np.float64)


def exact(a, b, what):
    d = (a - b).abs().max().item()
    assert d == 0.0, f"oracle restatement differs from the reference at {what}: max-abs {d}"


def gen_case(tag, batch, size, soft, seed, full_output):
    p = O.make_cunet_params(NC, seed)
    net = ref_cunet.Conditional_UNet(NC)
    missing = net.load_state_dict(p, strict=True)
    net.eval()
    x, c = O.make_inputs(batch, size, NC, seed, soft)
    hooks, stages = [], {}
    names = {"dconv_down1": "conv1", "dconv_down2": "conv2", "dconv_down3": "conv3",
             "dconv_down4": "bottleneck", "adain3": "adain3", "adain2": "adain2", "adain1": "adain1",
             "dconv_up3": "up3", "dconv_up2": "up2", "dconv_up1": "up1"}
    for mod, key in names.items():
        hooks.append(getattr(net, mod).register_forward_hook(
            lambda m, i, o, key=key: stages.__setitem__(key, o.detach().clone())))
    with torch.no_grad():
        y_ref = net(x, c)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        y_or, st = O.cunet_forward(p, x, c, None, return_stages=True)
    exact(y_ref, y_or, f"{tag} output")
    out = {"meta": np.array([batch, size, int(soft), seed, NC]), "out_summary": summary(y_ref)}
    for k, v in stages.items():
        exact(v, st[k], f"{tag} stage {k}")
        out["stage_" + k] = summary(v)
    if full_output:
        out["out"] = y_ref.numpy()

    # gradients of the benchmark loss mean|G(x,c) - x| (eval mode = dropout identity)
    net.zero_grad()
    loss = torch.mean(torch.abs(net(x, c) - x))
    loss.backward()
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_o = O.bench_loss(O.cunet_forward(pg, x, c), x)
    loss_o.backward()
    exact(loss.detach(), loss_o.detach(), f"{tag} loss")
    out["loss"] = np.array([loss.item()])
    for k, prm in net.named_parameters():
        if prm.grad is None:
            assert k.endswith("emb.weight")
            continue
        exact(prm.grad, pg[k].grad, f"{tag} grad {k}")
        out["grad_" + k] = summary(prm.grad)
    np.savez_compressed(os.path.join(OUT, f"cunet_{tag}.npz"), **out)
    print(f"cunet_{tag}: out abs-max {y_ref.abs().max():.4f} mean {y_ref.mean():.4f} loss {loss.item():.5f}")


def gen_train_case(tag, batch, size, seed):
    """Train mode (dropout active, as in the inference scripts that never call .eval(),
    inf_transfer_c.py:88-96).  nn.Dropout on CPU draws empty_like(x).bernoulli_(1-p) per call, so
    re-seeding and drawing the three masks in call order reproduces the reference's masks."""
    p = O.make_cunet_params(NC, seed)
    net = ref_cunet.Conditional_UNet(NC)
    net.load_state_dict(p, strict=True)
    net.train()
    x, c = O.make_inputs(batch, size, NC, seed, True)
    torch.manual_seed(1234)
    with torch.no_grad():
        y_ref = net(x, c)
    torch.manual_seed(1234)
    s = size // 4
    masks = []
    for ch, hw in ((512, s), (256, 2 * s), (128, 4 * s)):
        masks.append(torch.empty(batch, ch, hw, hw).bernoulli_(0.7))
    with torch.no_grad():
        y_or = O.cunet_forward(p, x, c, masks)
    exact(y_ref, y_or, f"{tag} train-mode output with reproduced masks")
    packed = [np.packbits(m.numpy().astype(np.uint8).reshape(-1)) for m in masks]
    np.savez_compressed(os.path.join(OUT, f"cunet_{tag}.npz"), meta=np.array([batch, size, 1, seed, NC]),
                        out=y_ref.numpy(), mask3=packed[0], mask2=packed[1], mask1=packed[2])
    print(f"cunet_{tag}: train-mode out abs-max {y_ref.abs().max():.4f}")


def gen_default_init_case(tag, batch, size, seed):
    """The literal reading of north_star's "on fixed seeds": torch.manual_seed(seed) + the modules' DEFAULT
    init (nn.Conv2d kaiming-uniform a=sqrt(5); cunet.py:41 leaves init_weight() commented out).  The build's
    modules consume the CPU RNG in the same order, so the GPU box regenerates identical weights from the seed;
    per-key checksums are stored so the test can prove that."""
    torch.manual_seed(seed)
    net = ref_cunet.Conditional_UNet(NC)
    net.eval()
    p = {k: v.clone() for k, v in net.state_dict().items()}
    x, c = O.make_inputs(batch, size, NC, seed, True)
    with torch.no_grad():
        y_ref = net(x, c)
        y_or = O.cunet_forward(p, x, c)
    exact(y_ref, y_or, f"{tag} default-init output")
    keys = sorted(p)
    chk = np.array([[p[k].double().sum().item(), p[k].double().abs().sum().item(), p[k].reshape(-1)[0].item()] for k in keys])
    np.savez_compressed(os.path.join(OUT, f"cunet_{tag}.npz"), meta=np.array([batch, size, 1, seed, NC]),
                        out=y_ref.numpy(), keys=np.array(keys), checksums=chk)
    print(f"cunet_{tag}: default-init out abs-max {y_ref.abs().max():.4f} std {y_ref.std():.5f}")


def gen_disc(tag, batch, size, seed):
    p = O.make_sndisc_params(NC, seed)
    net = ref_disc.SNDisc(NC)
    net.load_state_dict(p, strict=True)
    net.train()
    x, c = O.make_inputs(batch, size, NC, seed, True)
    outs = net(x, c)
    loss = torch.mean(torch.relu(1.0 - outs[0]))
    loss.backward()
    pg = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone())
          for k, v in p.items()}
    outs_o, nb = O.sndisc_forward(pg, x, c, train=True)
    loss_o = torch.mean(torch.relu(1.0 - outs_o[0]))
    loss_o.backward()
    out = {"meta": np.array([batch, size, 1, seed, NC]), "out": outs[0].detach().numpy(),
           "loss": np.array([loss.item()])}
    for i, (a, b) in enumerate(zip(outs, outs_o)):
        exact(a, b, f"{tag} D output {i}")
        out[f"o{i}_summary"] = summary(a)
    sd = net.state_dict()
    for k, v in nb.items():
        exact(sd[k], v, f"{tag} buffer {k}")
        out["buf_" + k] = v.numpy()
    for k, prm in net.named_parameters():
        exact(prm.grad, pg[k].grad, f"{tag} grad {k}")
        out["grad_" + k] = summary(prm.grad)
    # a second forward in eval mode (no power iteration) on the updated buffers
    net.eval()
    with torch.no_grad():
        outs_e = net(x, c)
    p2 = dict(p)
    p2.update(nb)
    with torch.no_grad():
        outs_eo, _ = O.sndisc_forward(p2, x, c, train=False)
    exact(outs_e[0], outs_eo[0], f"{tag} eval-mode D output")
    out["out_eval"] = outs_e[0].numpy()
    np.savez_compressed(os.path.join(OUT, f"sndisc_{tag}.npz"), **out)
    print(f"sndisc_{tag}: out {outs[0].detach().reshape(-1)[:4].numpy()} loss {loss.item():.5f}")


def gen_disc_default_init(tag, batch, size, seed):
    """torch.manual_seed(seed); SNDisc(nc): the reference's effective default init (disc.py:16-25: xavier_uniform_ through the
    storage-sharing `.weight` attribute), per-key checksums + one train-mode forward (output and moved buffers)."""
    torch.manual_seed(seed)
    net = ref_disc.SNDisc(NC)
    net.train()
    p = {k: v.clone() for k, v in net.state_dict().items()}
    x, c = O.make_inputs(batch, size, NC, seed, True)
    with torch.no_grad():
        outs = net(x, c)
        outs_o, nb = O.sndisc_forward(p, x, c, train=True)
    exact(outs[0], outs_o[0], f"{tag} default-init D output")
    sd = net.state_dict()
    for k, v in nb.items():
        exact(sd[k], v, f"{tag} buffer {k}")
    keys = sorted(p)
    chk = np.array([[p[k].double().sum().item(), p[k].double().abs().sum().item(), p[k].reshape(-1)[0].item(),
                     p[k].abs().max().item()] for k in keys])
    np.savez_compressed(os.path.join(OUT, f"sndisc_{tag}.npz"), meta=np.array([batch, size, 1, seed, NC]),
                        out=outs[0].numpy(), keys=np.array(keys), checksums=chk)
    print(f"sndisc_{tag}: default-init out {outs[0].reshape(-1)[:3].numpy()}")


def periodic_fill(shapes, seed, bounds):
    """Checkpoint-fixture weights: each tensor reads a 4093-entry table of U(-1,1) values cyclically from a key-dependent
    offset, times a per-tensor bound.  4093 is prime (no alignment with any tensor dimension, so a transposed / permuted load
    cannot go unnoticed) and 16 KiB of period keeps the 40-MB checkpoint a few hundred KB under gzip."""
    import zlib
    table = torch.from_numpy(np.random.default_rng([77, seed]).uniform(-1, 1, 4093).astype(np.float32))
    out = {}
    for k, shp in sorted(shapes.items()):
        n = int(np.prod(shp))
        idx = (torch.arange(n) + zlib.crc32(k.encode()) % 4093) % 4093
        out[k] = (table[idx] * bounds(k, shp)).reshape(shp).contiguous()
    return out


def gen_checkpoint(tag, batch, size, seed):
    """A checkpoint WRITTEN FROM THE REFERENCE MODULES' state_dict() in the format of t_est_train.py:365-373
    ({'inference', 'discriminator', 'epoch', 'global_step'}, tensors and ints only), gzip-compressed, plus what the
    reference modules compute from it: the per-class sweep of inference/inf_transfer_c.py:114-121, the signal-row sweep of
    inf_transfer_e.py:136-143, the direct per-row call of inf_1year_signals.py:104, D's outputs."""
    import gzip
    import io
    import math

    def g_bound(k, shp):
        if k.endswith("emb.weight"):
            return 1.0
        if k.endswith("weight"):
            fan_in = int(np.prod(shp[1:]))
            return 0.6 * (math.sqrt(2.0) if ".l1." not in k and k != "conv_last.weight" else 1.0) * math.sqrt(3.0 / fan_in)
        return 0.1

    def d_bound(k, shp):
        if k.endswith("weight_orig"):
            return math.sqrt(6.0 / int(np.prod(shp[1:])))
        return 1.0 if k.endswith(("weight_u", "weight_v")) else 0.1

    gp = periodic_fill(O.cunet_param_shapes(NC), seed, g_bound)
    dp = periodic_fill(O.sndisc_param_shapes(NC), seed, d_bound)
    for k in dp:
        if k.endswith(("weight_u", "weight_v")):
            dp[k] = torch.nn.functional.normalize(dp[k], dim=0, eps=1e-12)
    G = ref_cunet.Conditional_UNet(NC)
    G.load_state_dict(gp, strict=True)
    D = ref_disc.SNDisc(NC)
    D.load_state_dict(dp, strict=True)
    epoch, step = 3, 1234
    state = {"inference": G.state_dict(), "discriminator": D.state_dict(), "epoch": epoch, "global_step": step}   # :367-372
    buf = io.BytesIO()
    torch.save(state, buf)
    name = f"ref_ckpt_e{epoch:04d}_s{step}.pt"
    with gzip.GzipFile(os.path.join(OUT, name + ".gz"), "wb", compresslevel=9, mtime=0) as fh:
        fh.write(buf.getvalue())
    G.eval()
    D.eval()
    x, _ = O.make_inputs(batch, size, NC, seed, False)
    onehot = torch.eye(NC)
    rs = np.random.default_rng([78, seed])
    signals = torch.from_numpy(rs.standard_normal((3, NC)).astype(np.float32))      # standardised signal rows (t_est_train.py:131)
    per_row = torch.from_numpy(rs.standard_normal((batch, NC)).astype(np.float32))
    with torch.no_grad():
        cls = torch.stack([G(x, torch.cat([onehot[i]] * batch).view(-1, NC)) for i in range(NC)])        # inf_transfer_c.py:116-117
        sig = torch.stack([G(x, torch.cat([signals[i]] * batch).view(-1, NC)) for i in range(len(signals))])   # inf_transfer_e.py:138-139
        row = G(x, per_row)                                                                                   # inf_1year_signals.py:104
        d_out = D(cls[1], torch.cat([onehot[1]] * batch).view(-1, NC))[0]
        # the oracle on the same tensors
        exact(cls[2], O.cunet_forward(gp, x, torch.cat([onehot[2]] * batch).view(-1, NC)), f"{tag} sweep class 2")
        exact(row, O.cunet_forward(gp, x, per_row), f"{tag} per-row signals")
    np.savez_compressed(os.path.join(OUT, f"ref_ckpt_{tag}.npz"), meta=np.array([batch, size, 0, seed, NC, epoch, step]),
                        ckpt_file=np.array(name + ".gz"), class_sweep=cls.numpy(), signals=signals.numpy(), signal_sweep=sig.numpy(),
                        per_row=per_row.numpy(), per_row_out=row.numpy(), d_out=d_out.numpy())
    print(f"ref_ckpt_{tag}: {os.path.getsize(os.path.join(OUT, name + '.gz')) / 1024:.0f} KiB gzip of {len(buf.getvalue()) / 2**20:.1f} MiB; "
          f"sweep abs-max {cls.abs().max():.4f}, class-to-class max diff {(cls[0] - cls[1]).abs().max():.4f}")


def gen_eval_train_mode(tag, bs, size, seed):
    """evaluation() the way the reference runs it (t_cls_train.py:331-352): D in TRAIN mode -- 2*B power iterations under
    no_grad --, here with G in eval mode (its Dropout draws cannot be reproduced off the reference's RNG stream).  The loop
    body is the reference's, executed on the REFERENCE modules; stored: the four means, D's buffers after the sweep."""
    gp, dp = O.make_cunet_params(NC, seed), O.make_sndisc_params(NC, seed)
    G = ref_cunet.Conditional_UNet(NC)
    G.load_state_dict(gp, strict=True)
    G.eval()
    D = ref_disc.SNDisc(NC)
    D.load_state_dict(dp, strict=True)
    D.train()
    images, labels = O.make_inputs(bs, size, NC, seed, True)
    _, ref_labels = O.make_inputs(bs, size, NC, seed + 1, True)
    lin = torch.nn.Linear(3 * 8 * 8, NC)
    with torch.no_grad():
        lin.weight.copy_(O._uniform("evalfix.w", (NC, 192), 0.1, seed))
        lin.bias.copy_(O._uniform("evalfix.b", (NC,), 0.1, seed))
    est = lambda t: lin(torch.nn.functional.adaptive_avg_pool2d(t, 8).flatten(1))      # noqa: E731  (stand-in estimator_)
    adv, l1, w, d = [], [], [], []
    for i in range(bs):
        with torch.no_grad():
            ref_expand = torch.cat([ref_labels[i]] * bs).view(-1, NC)
            fake = G(images, ref_expand)
            fake_c = est(fake)
            real_d = D(images, labels)[0]
            fake_d = D(fake, ref_expand)[0]
        adv.append(ref_ops.gen_hinge(fake_d).item())
        l1.append(ref_ops.l1_loss(fake, images).item())
        w.append(ref_ops.pred_loss(fake_c, ref_expand).item())
        d.append(ref_ops.dis_hinge(fake_d, real_d).item())
    means = {"g_loss_adv": np.mean(adv), "g_loss_l1": np.mean(l1), "g_loss_w": np.mean(w), "d_loss": np.mean(d)}
    om, _, ob = O.evaluation(gp, dp, est, est, images, labels, ref_labels, d_train=True)
    for k, v in means.items():
        assert abs(om[k] - v) <= 1e-7 * max(1.0, abs(v)), (k, om[k], v)
    sd = D.state_dict()
    out = {"meta": np.array([bs, size, 1, seed, NC]), "means": np.array([means[k] for k in sorted(means)]),
           "mean_keys": np.array(sorted(means)), "est_w": lin.weight.detach().numpy(), "est_b": lin.bias.detach().numpy()}
    for k, v in ob.items():
        exact(sd[k], v, f"{tag} D buffer {k} after the sweep")
        out["buf_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"eval_{tag}.npz"), **out)
    print(f"eval_{tag}: {means}")


if __name__ == "__main__":
    assert not __debug__, "run with python3 -O (reference asserts CUDA tensors, utils.py:35)"
    cases = [
        (gen_case, ("c1_b2_128_onehot", 2, 128, False, 0, True)),      # BASELINE.json configs[0]
        (gen_case, ("b2_64_soft", 2, 64, True, 1, True)),
        (gen_case, ("b1_32_soft", 1, 32, True, 2, True)),
        (gen_case, ("b3_96x_onehot", 3, 96, False, 3, False)),
        (gen_train_case, ("train_b2_64", 2, 64, 4)),
        (gen_default_init_case, ("default_init_b2_128", 2, 128, 0)),
        (gen_default_init_case, ("default_init_b2_64", 2, 64, 5)),
        (gen_disc, ("b2_64", 2, 64, 0)),
        (gen_disc, ("b3_128", 3, 128, 1)),
        (gen_disc_default_init, ("default_init_b2_64", 2, 64, 7)),      # round 3
        (gen_checkpoint, ("b2_64", 2, 64, 11)),
        (gen_eval_train_mode, ("dtrain_b3_32", 3, 32, 8)),
    ]
    only = sys.argv[1:]            # e.g. `make_golden.py gen_checkpoint gen_eval` regenerates just those
    for fn, a in cases:
        if not only or any(o in fn.__name__ for o in only):
            fn(*a)
    print("golden vectors written to", OUT)
