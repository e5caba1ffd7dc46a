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


def test_sndisc_default_init_matches_reference(golden_dir):
    """torch.manual_seed(s); SNDisc(nc) reproduces the REFERENCE's effective default init (disc.py:16-25: xavier_uniform_
    reaches weight_orig through the storage-sharing `.weight` attribute spectral_norm registers) for all 40 state-dict
    tensors -- checksums captured from the reference --, and the oracle reproduces the reference's output on them."""
    import disc
    g = _load(golden_dir, "sndisc_default_init_b2_64.npz")
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    torch.manual_seed(seed)
    sd = disc.SNDisc(nc).state_dict()
    keys = [str(k) for k in g["keys"]]
    assert sorted(sd) == keys and len(keys) == 40
    for k, chk in zip(keys, g["checksums"]):
        v = sd[k].double()
        np.testing.assert_allclose([v.sum().item(), v.abs().sum().item(), sd[k].reshape(-1)[0].item(), sd[k].abs().max().item()],
                                   chk, rtol=1e-12, atol=1e-12)
    # the xavier bound (gain sqrt 2), not nn.Conv2d's default kaiming bound, is what the reference's weight_orig obeys
    w = sd["conv4.1.weight_orig"]
    assert w.abs().max().item() > 1.5 / np.sqrt(256 * 9)
    x, c = O.make_inputs(batch, size, nc, seed, True)
    with torch.no_grad():
        outs, _ = O.sndisc_forward({k: v.clone() for k, v in sd.items()}, x, c, train=True)
    assert np.abs(outs[0].numpy() - g["out"]).max() <= 1e-4 * max(1.0, float(np.abs(g["out"]).max()))


def _unpack_ckpt(golden_dir, tmp_path, g):
    import gzip
    name = str(g["ckpt_file"])
    path = os.path.join(str(tmp_path), "run", name[:-3])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.open(os.path.join(golden_dir, name), "rb") as src, open(path, "wb") as dst:
        dst.write(src.read())
    return path


def test_reference_written_checkpoint_loads_and_oracle_reproduces_the_sweeps(golden_dir, tmp_path):
    """a checkpoint written from the REFERENCE modules' state_dict() (format of t_est_train.py:365-373) loads
    into the build's modules with ``weights_only=True``; the oracle on the loaded tensors reproduces what the reference
    modules computed from the same file: the class sweep (inf_transfer_c.py:114-121), the signal-row sweep
    (inf_transfer_e.py:136-143), one row per image (inf_1year_signals.py:104), D's output."""
    import cunet
    import disc
    from wu.infer_driver import latest_checkpoint, load_checkpoint
    g = _load(golden_dir, "ref_ckpt_b2_64.npz")
    batch, size, soft, seed, nc, epoch, step = [int(v) for v in g["meta"]]
    path = _unpack_ckpt(golden_dir, tmp_path, g)
    assert latest_checkpoint(str(tmp_path), "run") == path
    G, D = cunet.Conditional_UNet(nc), disc.SNDisc(nc)
    assert load_checkpoint(path, G, D) == (epoch, step)
    gp = {k: v.clone() for k, v in G.state_dict().items()}
    dp = {k: v.clone() for k, v in D.state_dict().items()}
    assert set(gp) == set(O.cunet_param_shapes(nc)) and set(dp) == set(O.sndisc_param_shapes(nc))
    x, _ = O.make_inputs(batch, size, nc, seed, False)
    eye = torch.eye(nc)
    with torch.no_grad():
        for i in range(nc):
            out = O.cunet_forward(gp, x, eye[i].expand(batch, nc))
            assert np.abs(out.numpy() - g["class_sweep"][i]).max() <= TOL
        sig = torch.from_numpy(g["signals"])
        for i in range(sig.shape[0]):
            out = O.cunet_forward(gp, x, sig[i].expand(batch, nc))
            assert np.abs(out.numpy() - g["signal_sweep"][i]).max() <= TOL
        out = O.cunet_forward(gp, x, torch.from_numpy(g["per_row"]))
        assert np.abs(out.numpy() - g["per_row_out"]).max() <= TOL
        d_out = O.sndisc_forward(dp, torch.from_numpy(g["class_sweep"][1]), eye[1].expand(batch, nc), train=False)[0][0]
        assert np.abs(d_out.numpy() - g["d_out"]).max() <= 1e-4 * max(1.0, float(np.abs(g["d_out"]).max()))


def test_save_image_normalisation_formula():
    """wu.infer_driver.normalize_minmax / to_uint8 against the arithmetic of torchvision<0.4's save_image(normalize=True) on a
    single image (utils.make_grid.norm_ip: clamp_(min, max); add_(-min).div_(max - min + 1e-5); then mul(255).clamp(0, 255)
    .byte()), restated on CPU tensors.  torchvision itself is not importable here: the wrapper is restated, not pinned."""
    from wu.infer_driver import normalize_minmax, to_uint8
    r = np.random.default_rng(3)
    imgs = torch.from_numpy(r.uniform(-1, 1, (3, 3, 9, 11)).astype(np.float32))
    imgs[2] = 0.25                                                     # a constant image: (x - min) / 1e-5 = 0
    got = normalize_minmax(imgs)
    for j in range(3):
        img = imgs[j].clone()
        lo, hi = float(img.min()), float(img.max())
        img.clamp_(min=lo, max=hi)
        img.add_(-lo).div_(hi - lo + 1e-5)
        assert torch.equal(got[j], img.clamp(0, 1))
        assert torch.equal(to_uint8(got[j:j + 1])[0], img.mul(255).clamp(0, 255).byte().permute(1, 2, 0))


def test_evaluation_with_discriminator_in_train_mode(golden_dir):
    """evaluation() the way the reference runs it (t_cls_train.py:331-352): D in train mode, 2*B power iterations.  The fixture
    was computed by the reference's loop body on the REFERENCE modules; the oracle's ``d_train=True`` reproduces the four
    means and D's buffers after the sweep."""
    g = _load(golden_dir, "eval_dtrain_b3_32.npz")
    bs, size, soft, seed, nc = [int(v) for v in g["meta"]]
    gp, dp = O.make_cunet_params(nc, seed), O.make_sndisc_params(nc, seed)
    images, labels = O.make_inputs(bs, size, nc, seed, True)
    _, ref_labels = O.make_inputs(bs, size, nc, seed + 1, True)
    w, b = torch.from_numpy(g["est_w"]), torch.from_numpy(g["est_b"])
    est = lambda t: torch.nn.functional.linear(torch.nn.functional.adaptive_avg_pool2d(t, 8).flatten(1), w, b)   # noqa: E731
    means, _, bufs = O.evaluation(gp, dp, est, est, images, labels, ref_labels, d_train=True)
    for k, v in zip([str(k) for k in g["mean_keys"]], g["means"]):
        assert abs(means[k] - float(v)) <= 1e-5 * max(1.0, abs(float(v))), (k, means[k], v)
    moved = 0
    for k, v in bufs.items():
        assert np.abs(v.numpy() - g["buf_" + k]).max() <= TOL
        moved += int(not torch.equal(v, dp[k]))
    assert len(bufs) == 20 and moved >= 18          # every SN layer's u / v moved (2*B iterations)
    # and the eval-mode restatement leaves them alone and gives different d_loss (the deviation the train-mode path removes)
    means_e, _ = O.evaluation(gp, dp, est, est, images, labels, ref_labels)
    assert abs(means_e["d_loss"] - means["d_loss"]) > 0
