"""CPU: the oracle restatement (oracle/cunet_ref.py) against the golden vectors captured from the
reference itself (tests/golden/make_golden.py).  Tolerance 0 would hold on the generating machine;
a different CPU may pick other conv algorithms, so 1e-5 abs is allowed."""
import os

import numpy as np
import pytest
import torch

from oracle import cunet_ref as O

TOL = 1e-5


def _summary(t, nsamp=64):
    t = t.detach().reshape(-1).double()
    idx = torch.linspace(0, t.numel() - 1, nsamp).long()
    return np.concatenate([[t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()], t[idx].numpy()])


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("tag", ["c1_b2_128_onehot", "b2_64_soft", "b1_32_soft", "b3_96x_onehot"])
def test_cunet_forward_and_grads(golden_dir, tag):
    g = _load(golden_dir, f"cunet_{tag}.npz")
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    p = {k: v.requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    x, c = O.make_inputs(batch, size, nc, seed, bool(soft))
    out, st = O.cunet_forward(p, x, c, None, return_stages=True)
    if "out" in g:
        assert np.abs(out.detach().numpy() - g["out"]).max() <= TOL
    np.testing.assert_allclose(_summary(out), g["out_summary"], atol=TOL, rtol=0)
    for k, v in st.items():
        ref = g["stage_" + k]
        np.testing.assert_allclose(_summary(v), ref, atol=TOL * max(1.0, ref[1]), rtol=0)
    loss = O.bench_loss(out, x)
    assert abs(loss.item() - float(g["loss"][0])) <= TOL
    loss.backward()
    n = 0
    for k, prm in p.items():
        if k.endswith("emb.weight"):
            assert prm.grad is None        # utils.py:32 emb is never used in forward
            continue
        ref = g["grad_" + k]
        np.testing.assert_allclose(_summary(prm.grad), ref, atol=TOL * max(1.0, ref[1]), rtol=1e-4)
        n += 1
    assert n == 36


def test_cunet_train_mode_masks(golden_dir):
    g = _load(golden_dir, "cunet_train_b2_64.npz")
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    p = O.make_cunet_params(nc, seed)
    x, c = O.make_inputs(batch, size, nc, seed, bool(soft))
    s = size // 4
    masks = []
    for key, ch, hw in (("mask3", 512, s), ("mask2", 256, 2 * s), ("mask1", 128, 4 * s)):
        n = batch * ch * hw * hw
        masks.append(torch.from_numpy(np.unpackbits(g[key])[:n].reshape(batch, ch, hw, hw).astype(np.float32)))
    with torch.no_grad():
        out = O.cunet_forward(p, x, c, masks)
    assert np.abs(out.numpy() - g["out"]).max() <= TOL


@pytest.mark.parametrize("tag", ["b2_64", "b3_128"])
def test_sndisc(golden_dir, tag):
    g = _load(golden_dir, f"sndisc_{tag}.npz")
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    p = O.make_sndisc_params(nc, seed)
    for k in p:
        if k.endswith(("weight_orig", "bias")):
            p[k].requires_grad_(True)
    x, c = O.make_inputs(batch, size, nc, seed, True)
    outs, nb = O.sndisc_forward(p, x, c, train=True)
    scale = max(1.0, float(np.abs(g["out"]).max()))
    assert np.abs(outs[0].detach().numpy() - g["out"]).max() <= 1e-4 * scale
    for i, o in enumerate(outs):
        ref = g[f"o{i}_summary"]
        np.testing.assert_allclose(_summary(o), ref, atol=1e-4 * max(1.0, ref[1]), rtol=0)
    for k, v in nb.items():
        assert np.abs(v.numpy() - g["buf_" + k]).max() <= TOL
    loss = torch.mean(torch.relu(1.0 - outs[0]))
    assert abs(loss.item() - float(g["loss"][0])) <= 1e-4
    loss.backward()
    for k, prm in p.items():
        if prm.requires_grad:
            ref = g["grad_" + k]
            np.testing.assert_allclose(_summary(prm.grad), ref, atol=1e-4 * max(1.0, ref[1]), rtol=1e-3)
    p2 = {k: v.detach() for k, v in p.items()}
    p2.update(nb)
    with torch.no_grad():
        outs_e, _ = O.sndisc_forward(p2, x, c, train=False)
    assert np.abs(outs_e[0].numpy() - g["out_eval"]).max() <= 1e-4 * scale


@pytest.mark.parametrize("tag", ["default_init_b2_128", "default_init_b2_64"])
def test_default_init_matches_reference(golden_dir, tag):
    """torch.manual_seed(s) + the build's module constructor reproduces the REFERENCE's default-initialised
    weights (same RNG consumption order: checksums captured from the reference), and the oracle on those
    weights reproduces the reference's output."""
    import cunet
    g = _load(golden_dir, f"cunet_{tag}.npz")
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    torch.manual_seed(seed)
    sd = cunet.Conditional_UNet(nc).state_dict()
    keys = [str(k) for k in g["keys"]]
    assert sorted(sd) == keys
    for k, chk in zip(keys, g["checksums"]):
        v = sd[k].double()
        np.testing.assert_allclose([v.sum().item(), v.abs().sum().item(), sd[k].reshape(-1)[0].item()], chk, rtol=1e-12, atol=1e-12)
    x, c = O.make_inputs(batch, size, nc, seed, True)
    with torch.no_grad():
        out = O.cunet_forward(sd, x, c)
    assert np.abs(out.numpy() - g["out"]).max() <= TOL
