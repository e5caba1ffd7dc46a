"""GPU parity of the drop-in modules (cunet.Conditional_UNet, disc.SNDisc) against
 (a) the golden vectors captured from the reference itself (tests/golden/*.npz), and
 (b) the CPU oracle run in the same process on the same seeded inputs (stage-by-stage, gradients,
     train-mode dropout with the kernel's own mask).
Tolerances are the ones BASELINE.json's north_star states: forward max-abs <= 1e-3 (fp32), <= 5e-2 (bf16)."""
import os

import numpy as np
import pytest
import torch

from oracle import cunet_ref as O

pytestmark = pytest.mark.gpu
FWD_TOL = {"fp32": 1e-3, "bf16": 5e-2}
DEV = "cuda:0"


def _summary(t, nsamp=64):
    t = t.detach().reshape(-1).double().cpu()
    idx = torch.linspace(0, t.numel() - 1, nsamp).long()
    return np.concatenate([[t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()], t[idx].numpy()])


def _make_g(nc, seed, precision):
    import cunet
    net = cunet.Conditional_UNet(nc, precision=precision)
    net.load_state_dict(O.make_cunet_params(nc, seed), strict=True)
    return net.to(DEV)


def _make_d(nc, seed, precision):
    import disc
    net = disc.SNDisc(nc, precision=precision)
    net.load_state_dict(O.make_sndisc_params(nc, seed), strict=True)
    return net.to(DEV)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["c1_b2_128_onehot", "b2_64_soft", "b1_32_soft", "b3_96x_onehot"])
def test_cunet_forward_golden(golden_dir, tag, precision):
    g = np.load(os.path.join(golden_dir, f"cunet_{tag}.npz"))
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    net = _make_g(nc, seed, precision).eval()
    x, c = O.make_inputs(batch, size, nc, seed, bool(soft))
    with torch.no_grad():
        out = net(x.to(DEV), c.to(DEV))
    assert out.dtype == torch.float32 and tuple(out.shape) == (batch, 3, size, size) and out.is_contiguous()
    if "out" in g:
        err = np.abs(out.cpu().numpy() - g["out"]).max()
        print(f"{tag} {precision}: max-abs vs reference {err:.3e}")
        assert err <= FWD_TOL[precision], f"max-abs vs reference {err}"
    s = _summary(out)
    assert np.abs(s[3:] - g["out_summary"][3:]).max() <= FWD_TOL[precision]
    assert abs(s[0] - g["out_summary"][0]) <= FWD_TOL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["default_init_b2_128", "default_init_b2_64"])
def test_cunet_forward_default_init_golden(golden_dir, tag, precision):
    """north_star's literal case: torch.manual_seed + DEFAULT init (cunet.py:41 leaves init_weight() off); the
    module constructor draws the same weights as the reference's (proved by checksums in the CPU suite)."""
    import cunet
    g = np.load(os.path.join(golden_dir, f"cunet_{tag}.npz"))
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    torch.manual_seed(seed)
    net = cunet.Conditional_UNet(nc, precision=precision).to(DEV).eval()
    x, c = O.make_inputs(batch, size, nc, seed, True)
    with torch.no_grad():
        out = net(x.to(DEV), c.to(DEV))
    err = np.abs(out.cpu().numpy() - g["out"]).max()
    print(f"default-init {tag} {precision}: max-abs vs reference {err:.3e}")
    assert err <= FWD_TOL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cunet_gradients_golden(golden_dir, precision):
    """Gradients of the benchmark loss mean|G(x,c)-x| against the reference's autograd (golden samples)."""
    g = np.load(os.path.join(golden_dir, "cunet_b2_64_soft.npz"))
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    net = _make_g(nc, seed, precision).eval()
    x, c = O.make_inputs(batch, size, nc, seed, bool(soft))
    xd = x.to(DEV)
    loss = torch.mean(torch.abs(net(xd, c.to(DEV)) - xd))
    loss.backward()
    assert abs(loss.item() - float(g["loss"][0])) <= FWD_TOL[precision]
    # 64 strided samples of every parameter gradient, captured from the reference's autograd.
    # fp32: sample error relative to the gradient's rms.  The gradient is a DISCONTINUOUS function of the
    # forward values: a single ReLU gate (or max-pool arg-max) flipped by a 1e-7 forward difference changes an
    # activation gradient of N live elements by ~1/sqrt(N) relative (2e-3 at 64x64x64, 6e-3 at the 8x8x512
    # bottleneck) -- the CPU oracle itself differs from a float64 run of the same code by 2-3e-3 per layer.
    # The kernels themselves are checked to ~3e-7 in test_gpu_kernels.py.  bf16: the backward chain rounds the
    # activation gradient to bf16 at each of ~30 ops, so the check is the direction of the sample vector.
    worst = 0.0
    for k, prm in net.named_parameters():
        if k.endswith("emb.weight"):
            assert prm.grad is None
            continue
        ref = g["grad_" + k]
        got = _summary(prm.grad)
        rms = max(ref[2], 1e-12)
        if precision == "fp32":
            err = np.abs(got[3:] - ref[3:]).max() / rms
            worst = max(worst, err)
            assert err <= 0.1, f"{k}: sample err/rms {err}"
            assert abs(got[2] - ref[2]) / rms <= 2e-2, f"{k}: rms {got[2]} vs {ref[2]}"
        else:
            a, b = got[3:], ref[3:]
            cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            worst = max(worst, 1 - cos)
            # vs the fp32 REFERENCE: a different precision mode.  Downstream of the last AdaIN nothing amplifies the bf16 rounding
            # (measured full-tensor cosine 0.9988+ there, 0.94-0.98 on the deep layers, where two CPU emulations of the bf16
            # graph agree no better: tests/test_host_cpu.py::test_bf16_sum_order_witness).  The kernels' own backward arithmetic
            # is pinned to cos >= 0.99999 stage by stage in test_gpu_round3.py::test_bf16_backward_stage_by_stage_vs_emulating_oracle.
            tight = k.startswith(("dconv_up1", "conv_last"))
            assert cos >= (0.99 if tight else 0.9), f"{k}: sample cosine {cos}"
            assert abs(got[2] - ref[2]) / rms <= (0.05 if tight else 0.15), f"{k}: rms {got[2]} vs {ref[2]}"
    print("worst grad sample deviation", worst)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cunet_stages_and_grads_vs_oracle(precision):
    """Full-tensor comparison with the CPU oracle on a fresh seeded case (B=2, 48x40: ragged tiles)."""
    nc, seed = 5, 7
    net = _make_g(nc, seed, precision).eval()
    p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    r = O._rng("ragged", seed)
    x = torch.from_numpy(r.uniform(-1, 1, size=(2, 3, 48, 40)).astype(np.float32))
    c = torch.softmax(torch.from_numpy(r.standard_normal((2, nc)).astype(np.float32)), 1)
    ref = O.cunet_forward(p, x, c)
    xd = x.to(DEV)
    out = net(xd, c.to(DEV))
    err = (out.detach().cpu() - ref.detach()).abs().max().item()
    assert err <= FWD_TOL[precision], f"forward {err}"
    O.bench_loss(ref, x).backward()
    torch.mean(torch.abs(out - xd)).backward()
    for k, prm in net.named_parameters():
        if k.endswith("emb.weight"):
            continue
        a, b = prm.grad.detach().cpu().reshape(-1).double(), p[k].grad.reshape(-1).double()
        cos = torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)
        rel = (a - b).norm() / (b.norm() + 1e-30)
        if precision == "fp32":
            lim = (0.999, 5e-2)
        elif k.startswith(("dconv_up1", "conv_last")):
            lim = (0.995, 0.1)          # bf16 vs the fp32 oracle, no AdaIN backward upstream (measured 0.9988+)
        else:
            lim = (0.95, 0.35)          # the precision mode's own spread on the deep layers (see test_cunet_gradients_golden)
        assert cos >= lim[0] and rel <= lim[1], f"{k}: cos {cos:.6f} rel {rel:.4f}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cunet_fused_graph_equals_per_layer_functions(precision):
    """The single-node fused schedule (wu/unet_graph.py: epilogue-gated gradients, fused skip-sum) and the
    per-layer autograd Functions are two schedules of the same kernels: same forward bits, same gradients up to the
    one place they round differently (the encoder skip-gradient sum)."""
    nc, seed = 5, 6
    x, c = O.make_inputs(2, 64, nc, seed, True)
    res = []
    for fused in (True, False):
        net = _make_g(nc, seed, precision).train()
        net.dropout_seed = 7
        net.fused = fused
        xd = x.to(DEV)
        out = net(xd, c.to(DEV))
        torch.mean(torch.abs(out - xd)).backward()
        res.append((out.detach(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    if precision == "fp32":
        assert torch.equal(res[0][0], res[1][0])
    else:
        # round 4: the fused graph computes the 64 -> 3 head + tanh in the last conv's epilogue on the matrix cores (same stored bf16
        # activations, fp32-exact products, another summation order than the stand-alone head kernel of the per-layer path)
        assert (res[0][0] - res[1][0]).abs().max().item() < 2e-6
    assert set(res[0][1]) == set(res[1][1]) and len(res[0][1]) == 36
    for k in res[0][1]:
        a, b = res[0][1][k].double().reshape(-1), res[1][1][k].double().reshape(-1)
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        assert rel <= (1e-5 if precision == "fp32" else 3e-2), f"{k}: {rel}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cunet_train_mode_dropout(precision):
    """Train mode (Dropout(0.3) active, as in inf_transfer_c.py which never calls .eval()): the oracle is
    given the kernel's own keep-masks (regenerated from the seed through the C ABI)."""
    from wu import functional as WF
    nc, seed, n, size = 5, 4, 2, 64
    net = _make_g(nc, seed, precision).train()
    net.dropout_seed = 99
    x, c = O.make_inputs(n, size, nc, seed, True)
    with torch.no_grad():
        out = net(x.to(DEV), c.to(DEV))
        out2 = net(x.to(DEV), c.to(DEV))
    assert torch.equal(out, out2)                     # fixed seed -> same masks
    s = size // 4
    masks = [WF.dropout_mask(n, ch, hw, hw, 0.3, (99 * 4 + k) & 0x7FFFFFFFFFFFFFFF, torch.device(DEV)).float().cpu()
             for k, ch, hw in ((3, 512, s), (2, 256, 2 * s), (1, 128, 4 * s))]
    ref = O.cunet_forward(O.make_cunet_params(nc, seed), x, c, masks)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= FWD_TOL[precision], f"train-mode forward {err}"
    net.dropout_seed = None
    with torch.no_grad():
        out3 = net(x.to(DEV), c.to(DEV))
    assert not torch.equal(out, out3)                 # fresh seed per call


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["b2_64", "b3_128"])
def test_sndisc_golden(golden_dir, tag, precision):
    g = np.load(os.path.join(golden_dir, f"sndisc_{tag}.npz"))
    batch, size, soft, seed, nc = [int(v) for v in g["meta"]]
    net = _make_d(nc, seed, precision).train()
    x, c = O.make_inputs(batch, size, nc, seed, True)
    outs = net(x.to(DEV), c.to(DEV))
    assert len(outs) == 5 and tuple(outs[0].shape) == (batch, 1)
    scale = max(1.0, float(np.abs(g["out"]).max()))
    tol = (2e-3 if precision == "fp32" else 5e-2) * scale
    assert np.abs(outs[0].detach().cpu().numpy() - g["out"]).max() <= tol
    for i in range(1, 5):
        ref = g[f"o{i}_summary"]
        got = _summary(outs[i].float())
        assert np.abs(got[3:] - ref[3:]).max() <= (1e-3 if precision == "fp32" else 3e-2) * max(1.0, ref[1])
    sd = net.state_dict()
    for k in g.files:
        if k.startswith("buf_"):                      # power-iteration buffers (fp32 GEMVs): must match tightly
            assert np.abs(sd[k[4:]].cpu().numpy() - g[k]).max() <= 1e-4, k
    loss = torch.mean(torch.relu(1.0 - outs[0]))
    loss.backward()
    for k, prm in net.named_parameters():
        ref = g["grad_" + k]
        got = _summary(prm.grad)
        rms = max(ref[2], 1e-12)
        err = np.abs(got[3:] - ref[3:]).max() / rms
        a, b = got[3:], ref[3:]
        cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        # bf16 vs the fp32 reference samples; the tight bf16 pin (all gradients cos >= 0.999 vs the bf16-emulating oracle) is
        # test_gpu_round3.py::test_sndisc_bf16_gradients_vs_emulating_oracle
        assert err <= (3e-2 if precision == "fp32" else 0.25) and cos >= (0.9999 if precision == "fp32" else 0.995), f"{k}: err/rms {err} cos {cos}"
    net.eval()
    with torch.no_grad():
        oe = net(x.to(DEV), c.to(DEV))[0]
    assert np.abs(oe.cpu().numpy() - g["out_eval"]).max() <= tol


def test_sndisc_input_gradient():
    """g_loss differentiates D wrt G's output (t_cls_train.py:243,272): d out / d x through all of D."""
    nc, seed = 5, 3
    net = _make_d(nc, seed, "fp32").eval()
    p = O.make_sndisc_params(nc, seed)
    x, c = O.make_inputs(2, 32, nc, seed, True)
    xr = x.clone().requires_grad_(True)
    outs, _ = O.sndisc_forward(p, xr, c, train=False)
    outs[0].sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    net(xd, c.to(DEV))[0].sum().backward()
    a, b = xd.grad.cpu().reshape(-1).double(), xr.grad.reshape(-1).double()
    assert (a - b).norm() / b.norm() <= 2e-3


def test_state_dict_round_trip_and_optimizer():
    """Checkpoint interchange: reference-keyed state-dicts load/save unchanged; Adam steps the
    fp32 OIHW parameters and the packed MFMA operands follow (repacked on version change)."""
    import cunet
    nc = 5
    net = _make_g(nc, 0, "bf16").eval()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = O.make_cunet_params(nc, 0)
    assert set(sd) == set(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
    x, c = O.make_inputs(2, 32, nc, 0, False)
    xd, cd = x.to(DEV), c.to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.0, 0.999), weight_decay=1e-3 / 20)   # t_cls_train.py:184
    with torch.no_grad():
        y0 = net(xd, cd).clone()
    for _ in range(2):
        opt.zero_grad()
        torch.mean(torch.abs(net(xd, cd) - xd)).backward()
        opt.step()
    with torch.no_grad():
        y1 = net(xd, cd)
    assert (y1 - y0).abs().max().item() > 1e-3        # the update reached the kernels' packed weights
    net2 = cunet.Conditional_UNet(nc).to(DEV).eval()
    net2.load_state_dict(net.state_dict())
    with torch.no_grad():
        assert torch.equal(net2(xd, cd), y1)


def test_hipgraph_inference_equals_eager():
    """configs[4]-style inference: the eval-mode forward captured into a hipGraph replays bit-identically to the
    eager forward, for new inputs copied into the static buffers, and follows weight updates after a re-pack."""
    from wu.graph_infer import GraphedUNet
    nc, seed = 5, 8
    net = _make_g(nc, seed, "bf16").eval()
    g = GraphedUNet(net, batch=2, size=64)
    for s in (1, 2):
        x, c = O.make_inputs(2, 64, nc, s, bool(s & 1))
        xd, cd = x.to(DEV), c.to(DEV)
        with torch.no_grad():
            eager = net(xd, cd)
        out = g(xd, cd, copy_out=True)
        assert torch.equal(out, eager)
    # (train-mode = dropout-active graphs: tests/test_gpu_round2.py::test_hipgraph_dropout_active_and_recapture)


def _gan_case(mode, precision, supervised=False, cross_ent=False, size=64, batch=2, seed=9):
    """One D update + one G update of the reference loop against the oracle's restatement (oracle/cunet_ref.py
    update_discriminator_loss / update_inference_loss) with the same weights, inputs and stand-in estimator; lr = 0 so the G
    update sees the D weights the oracle used.  Returns the worst G / D gradient cosines and the loss errors."""
    from wu.train_step import WeatherTransferStep, StandInEstimator
    nc = 5
    st = WeatherTransferStep(nc, mode=mode, precision=precision, device=DEV, ddp=False, seed=1, supervised=supervised, cross_ent=cross_ent)
    st.inference.load_state_dict(O.make_cunet_params(nc, seed))
    st.discriminator.load_state_dict(O.make_sndisc_params(nc, seed))
    st.inference.eval()                        # dropout identity so the oracle needs no mask
    x, _ = O.make_inputs(batch, size, nc, seed, True)
    xr, _ = O.make_inputs(batch, size, nc, seed + 1, True)
    c_d = torch.arange(batch) % nc
    c_r = (torch.arange(batch) * 2 + 1) % nc
    est_cpu = StandInEstimator(nc)
    est_cpu.load_state_dict({k: v.cpu() for k, v in st.estimator_.state_dict().items()})
    est_raw = est_cpu
    est_out = (lambda t: torch.softmax(est_cpu(t), 1)) if mode == "cls" else est_cpu      # t_cls_train.py:174-178
    # ---- oracle ----
    gp = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    dp = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone())
          for k, v in O.make_sndisc_params(nc, seed).items()}
    eye = torch.eye(nc)
    with torch.no_grad():
        if supervised:
            rand_labels, d_labels, r_idx = eye[c_r], eye[c_d], c_r                        # t_cls_train.py:419-421,432
        else:
            raw = est_raw(xr)
            rand_labels = est_out(xr)                                                    # :423
            d_labels, r_idx = None, (torch.argmax(raw, 1) if cross_ent else None)        # :436
    d_loss_ref, nb2 = O.update_discriminator_loss(gp, dp, est_out, x, rand_labels, d_labels, supervised)
    d_loss_ref.backward()
    d_grads = {k: v.grad.clone() for k, v in dp.items() if v.requires_grad}
    dp3 = {k: v.detach() for k, v in dp.items()}
    dp3.update(nb2)                                                                       # D's buffers after two power iterations
    g_ref = O.update_inference_loss(gp, dp3, est_out, est_raw, x, rand_labels, d_labels, r_idx, supervised, cross_ent)
    g_ref[0].backward()
    # ---- build ----
    for opt in (st.d_opt, st.g_opt):
        for g in opt.param_groups:
            g["lr"] = 0.0
            g["weight_decay"] = 0.0
    xd, xrd = x.to(DEV), xr.to(DEV)
    rl = rand_labels.to(DEV)
    dl = d_labels.to(DEV) if d_labels is not None else None
    d_loss = st.update_discriminator(xd, rl, dl)
    res = {"d_loss_err": abs(d_loss.item() - d_loss_ref.item()) / max(1.0, abs(d_loss_ref.item()))}
    worst_d = 1.0
    for k, prm in st.discriminator.named_parameters():
        a, b = prm.grad.detach().cpu().reshape(-1).double(), d_grads[k].reshape(-1).double()
        worst_d = min(worst_d, (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item())
    g_losses = st.update_inference(xd, rl, dl, r_idx.to(DEV) if r_idx is not None else None)
    res["g_loss_err"] = abs(g_losses[0].item() - g_ref[0].item()) / max(1.0, abs(g_ref[0].item()))
    res["g_parts_err"] = max(abs(g_losses[i].item() - g_ref[i].item()) / max(1.0, abs(g_ref[i].item())) for i in (1, 2, 3))
    worst_g = 1.0
    per_layer = {}
    for k, prm in st.inference.named_parameters():
        if prm.grad is None:
            continue
        a, b = prm.grad.detach().cpu().reshape(-1).double(), gp[k].grad.reshape(-1).double()
        per_layer[k] = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
        worst_g = min(worst_g, per_layer[k])
    res.update(worst_d=worst_d, worst_g=worst_g, per_layer=per_layer)
    return res


@pytest.mark.parametrize("mode,supervised,cross_ent", [("cls", False, False), ("est", False, False),
                                                       ("cls", True, False), ("cls", False, True), ("cls", True, True)])
def test_gan_step_matches_oracle(mode, supervised, cross_ent):
    """fp32: the t_cls_train step (t_cls_train.py:226-312) incl. its --supervised (:232-235,260-262,294-297) and --cross_ent
    (:247-251,256,436) switches, and the t_est_train soft-label step (t_est_train.py:214-283, BASELINE configs[3]: raw
    5-signal estimator outputs as conditioning, MSE weather loss) against the CPU oracle's autograd."""
    r = _gan_case(mode, "fp32", supervised, cross_ent)
    print(f"GAN step {mode} sup={supervised} ce={cross_ent}: d_loss err {r['d_loss_err']:.2e}, g_loss err {r['g_loss_err']:.2e}, "
          f"worst D cos {r['worst_d']:.8f}, worst G cos {r['worst_g']:.8f}")
    assert r["d_loss_err"] <= 2e-3 and r["g_loss_err"] <= 2e-3 and r["g_parts_err"] <= 2e-3
    assert r["worst_d"] >= 0.999 and r["worst_g"] >= 0.995


@pytest.mark.parametrize("mode,size", [("cls", 64), ("est", 64), ("cls", 256)])
def test_gan_step_bf16(mode, size):
    """The same two steps in the bf16 production mode (configs[2] / configs[3] arithmetic).  Tolerances: losses within 5e-2
    relative (north_star's bf16 forward tolerance; the hinge / reconstruction terms are means of O(1) values), D gradients
    cosine >= 0.995 (measured 0.9976-0.9983; D has no normalisation layer), G gradients cosine >= 0.94 (measured 0.959 / 0.978 /
    0.980: they cross D, the estimator and the three AdaIN boundaries, whose instance-norm backward amplifies one-ulp bf16 flips --
    two bf16 emulations that differ only in fp32 summation order already disagree by 0.951 on those layers,
    tests/test_host_cpu.py::test_bf16_sum_order_witness; the gradients downstream of the last AdaIN are at 0.9999+).  The per-layer
    figures are printed; the tight bf16 checks of the kernels' own backward arithmetic are the stage-by-stage tests
    (test_bf16_backward_stage_by_stage_vs_emulating_oracle: cos >= 0.99999; test_estimator_bf16_backward_block_by_block: >= 0.9999)."""
    r = _gan_case(mode, "bf16", size=size)            # size 256 = BASELINE configs[2]'s resolution (B=2: the CPU oracle's budget)
    print(f"bf16 GAN step {mode} {size}x{size}: d_loss err {r['d_loss_err']:.2e}, g_loss err {r['g_loss_err']:.2e}, worst D cos {r['worst_d']:.5f}, "
          f"worst G cos {r['worst_g']:.5f}")
    for k, v in r["per_layer"].items():
        print(f"   {k:24s} cos {v:.5f}")
    assert r["d_loss_err"] <= 5e-2 and r["g_loss_err"] <= 5e-2
    assert r["worst_d"] >= 0.995 and r["worst_g"] >= 0.94
    tail = [v for k, v in r["per_layer"].items() if k.startswith(("dconv_up1", "conv_last"))]
    assert tail and min(tail) >= 0.999, "gradients downstream of the last AdaIN boundary"


def test_checkpoint_interchange_and_class_sweep(tmp_path):
    """reference-format checkpoints ({'inference','discriminator','epoch','global_step'}) round-trip
    through save/load/resume, and the inf_transfer_c-style one-hot sweep equals per-class forwards."""
    import cunet
    import disc
    from wu.graph_infer import GraphedUNet
    from wu.infer_driver import class_sweep, latest_checkpoint, load_checkpoint, normalize_minmax, save_checkpoint
    nc = 5
    g, d = _make_g(nc, 11, "bf16").eval(), _make_d(nc, 11, "bf16")
    save_checkpoint(str(tmp_path), "run", g, d, 3, 2000)
    save_checkpoint(str(tmp_path), "run", g, d, 3, 3000)
    last = latest_checkpoint(str(tmp_path), "run")
    assert last.endswith("run_e0003_s3000.pt")
    raw = torch.load(last, weights_only=True)
    assert set(raw) == {"inference", "discriminator", "epoch", "global_step"}
    assert set(raw["inference"]) == set(O.cunet_param_shapes(nc)) and set(raw["discriminator"]) == set(O.sndisc_param_shapes(nc))
    g2, d2 = cunet.Conditional_UNet(nc).to(DEV).eval(), disc.SNDisc(nc).to(DEV)
    assert load_checkpoint(last, g2, d2) == (3, 3000)
    x, _ = O.make_inputs(2, 64, nc, 11, False)
    xd = x.to(DEV)
    sweep = class_sweep(g2, xd)
    assert tuple(sweep.shape) == (nc, 2, 3, 64, 64)
    eye = torch.eye(nc, device=DEV)
    for i in range(nc):
        with torch.no_grad():
            ref = g(xd, eye[i].repeat(2, 1))
        assert torch.equal(sweep[i], ref)
    assert torch.equal(class_sweep(g2, xd, graphed=GraphedUNet(g2, 2, 64)), sweep)
    nm = normalize_minmax(sweep[0])
    assert nm.min().item() >= 0 and abs(nm.reshape(2, -1).max(dim=1).values - 1).max().item() < 1e-3


def test_full_size_properties():
    """BASELINE configs[1] size (256x256, B=32, bf16), where the CPU oracle would take minutes: size-independent
    properties instead.  (a) run-to-run determinism of forward AND every parameter gradient (deterministic split-K,
    fixed-order reductions, seeded dropout); (b) batch independence: an image's output does not depend on what else is
    in the batch (to bf16 rounding); (c) the bf16 production kernels (LDS-DMA conv / wgrad) against the fp32 generic kernels on
    the same weights at full resolution: forward within the bf16 tolerance, gradients aligned."""
    nc = 5
    torch.manual_seed(3)
    net = _make_g(nc, 2, "bf16").train()
    net.dropout_seed = 5
    g = torch.Generator().manual_seed(11)
    x = (torch.rand((32, 3, 256, 256), generator=g) * 2 - 1).to(DEV)
    c = torch.eye(nc)[torch.arange(32) % nc].to(DEV)

    def run(model, xx, cc):
        for p in model.parameters():
            p.grad = None
        out = model(xx, cc)
        torch.mean(torch.abs(out - xx)).backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    o1, g1 = run(net, x, c)
    o2, g2 = run(net, x, c)
    assert torch.equal(o1, o2)
    for k in g1:        # every gradient, the thin 3-channel layers included (per-workgroup partials folded in a fixed order)
        assert torch.equal(g1[k], g2[k]), f"gradient of {k} is not reproducible"
    # (b) eval mode: images 6..7 alone == images 6..7 inside the batch of 32
    net.eval()
    with torch.no_grad():
        full = net(x, c)
        part = net(x[6:8].contiguous(), c[6:8].contiguous())
    # not bitwise: the instance-norm statistics split their H*W reduction by a factor chosen from N*C (to fill the chip), so the
    # fp32 summation order -- and hence an occasional bf16 rounding downstream -- depends on the batch size
    assert (full[6:8] - part).abs().max().item() <= 2e-2
    assert (full[6:8] - part).abs().mean().item() <= 1e-3
    # (c) fp32 generic kernels vs bf16 production kernels, B=4 at 256x256
    net32 = _make_g(nc, 2, "fp32").eval()
    xs, cs = x[:4].contiguous(), c[:4].contiguous()
    o32, g32 = run(net32, xs, cs)
    o16, g16 = run(net, xs, cs)
    assert (o32 - o16).abs().max().item() <= FWD_TOL["bf16"]
    # ALL 36 gradients (weights and biases of the 15 convs, the three AdaIN style layers), per-layer cosine printed.  The fp32
    # run is a different precision mode of the same kernels' algorithms, so ReLU / max-pool decisions flipped by bf16 rounding
    # bound the cosine from above (the tight bf16 check is test_gpu_round2.py::test_bf16_gradients_vs_emulating_oracle): decoder
    # and head >= 0.97, encoder (whose gradients cross ~25 gated ops) >= 0.9.
    assert len(g16) == 36 and set(g16) == set(g32)
    for k in g16:
        a, b = g16[k].double().reshape(-1), g32[k].double().reshape(-1)
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
        print(f"   256x256 bf16 vs fp32 kernels  {k:24s} cos {cos:.5f}")
        lim = 0.97 if k.startswith(("dconv_up", "conv_last")) else 0.9
        assert cos >= lim, f"{k}: cosine {cos}"


def test_fused_backward_gradient_sink():
    """wu.ddp.GradBucketReducer.attach(): the fused node accumulates dW / db straight into the reducer's bucket views and
    announces them layer by layer (the all-reduce of a full bucket then overlaps the rest of backward).  Single process:
    the collectives are no-ops, the gradients must equal the ordinary autograd path bit for bit, buckets must have been
    launched (in order) before backward returned, and a second backward without zero_grad must ACCUMULATE."""
    from wu.ddp import GradBucketReducer, ready_order
    nc = 5
    net = _make_g(nc, 4, "bf16").train()
    net.dropout_seed = 9
    x, c = (t.to(DEV) for t in O.make_inputs(2, 64, nc, 4, True))

    def run():
        out = net(x, c)
        torch.mean(torch.abs(out - x)).backward()

    for p in net.parameters():
        p.grad = None
    run()
    ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    red = GradBucketReducer(ready_order(net), bucket_mb=2.0, ready_order=True).attach(net)
    red.zero_grad()
    run()
    log = list(red.launch_log)
    red.finalize()
    assert len(red.buckets) >= 4 and log == list(range(len(log))) and len(log) >= len(red.buckets) - 1
    for k, p in net.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k
    run()                                                # no zero_grad: accumulate
    red.finalize()
    for k in ("dconv_up1.2.weight", "dconv_down4.0.bias", "adain2.fc_std.weight" if "adain2.fc_std.weight" in ref else "dconv_up3.0.weight"):
        assert torch.allclose(dict(net.named_parameters())[k].grad, 2 * ref[k], rtol=1e-5, atol=1e-8), k
    net.grad_sink = None


def test_side_stream_wgrad_is_bitwise_neutral():
    """The weight-gradient kernels run on a second HIP stream in the fused backward (wu/unet_graph.py): same kernels, same
    operands, different queue -- every gradient must be bit-identical to the single-stream schedule."""
    from wu import unet_graph as UG
    nc = 5
    net = _make_g(nc, 6, "bf16").train()
    net.dropout_seed = 3
    x, c = (t.to(DEV) for t in O.make_inputs(2, 64, nc, 6, True))
    grads = {}
    try:
        for flag in (True, False):
            UG.SIDE_STREAM_WGRAD = flag
            for p in net.parameters():
                p.grad = None
            torch.mean(torch.abs(net(x, c) - x)).backward()
            torch.cuda.synchronize()
            grads[flag] = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    finally:
        UG.SIDE_STREAM_WGRAD = True
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k


@pytest.mark.parametrize("mode", ["cls", "est"])
def test_step_shares_the_estimator_forwards(mode):
    """``step()`` runs the frozen estimator once on cat(rand_images, images) instead of the reference's three no-grad calls
    (t_cls_train.py:424, :297, :237 -- the last two on the same batch): one iteration must give the losses of the call-by-call
    sequence (fp32; two identically seeded harnesses)."""
    from wu.train_step import WeatherTransferStep
    nc, seed = 5, 9
    x, _ = O.make_inputs(2, 64, nc, seed, True)
    xr, _ = O.make_inputs(2, 64, nc, seed + 1, True)
    xd, xrd = x.to(DEV), xr.to(DEV)
    res = []
    for shared in (True, False):
        st = WeatherTransferStep(nc, mode=mode, precision="fp32", device=DEV, ddp=False, seed=3)
        st.inference.load_state_dict(O.make_cunet_params(nc, seed))
        st.discriminator.load_state_dict(O.make_sndisc_params(nc, seed))
        st.inference.dropout_seed = 4
        if shared:
            out = st.step(xd, xrd)
        else:
            with torch.no_grad():
                rl = st.estimator(xrd)
            out = (st.update_discriminator(xd, rl),) + st.update_inference(xd, rl)
        res.append([t.item() for t in out])
    for a, b in zip(*res):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), res


def test_gan_two_iterations_match_oracle():
    """Two FULL iterations (D update + G update, real Adam steps, lr 1e-3) of the t_cls_train loop against the oracle driven by
    stock torch.optim.Adam on the CPU: the second iteration's losses depend on every weight the first one wrote -- G's packed conv
    operands, D's spectral-norm weights and power-iteration buffers -- so a stale cache anywhere shows up here (the single-step
    test above runs with lr = 0 and cannot see it)."""
    from wu.train_step import WeatherTransferStep, StandInEstimator
    nc, seed, batch, size, lr = 5, 9, 2, 64, 1e-3
    st = WeatherTransferStep(nc, mode="cls", precision="fp32", lr=lr, device=DEV, ddp=False, seed=1)
    st.inference.load_state_dict(O.make_cunet_params(nc, seed))
    st.discriminator.load_state_dict(O.make_sndisc_params(nc, seed))
    st.inference.eval()                        # dropout identity so the oracle needs no mask
    x, _ = O.make_inputs(batch, size, nc, seed, True)
    xr, _ = O.make_inputs(batch, size, nc, seed + 1, True)
    est_cpu = StandInEstimator(nc)
    est_cpu.load_state_dict({k: v.cpu() for k, v in st.estimator_.state_dict().items()})
    est_out = lambda t: torch.softmax(est_cpu(t), 1)
    gp = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    dp = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone())
          for k, v in O.make_sndisc_params(nc, seed).items()}
    wd = st.g_opt.param_groups[0]["weight_decay"]
    g_ref_opt = torch.optim.Adam([v for v in gp.values()], lr=lr, betas=(0.0, 0.999), weight_decay=wd)
    d_ref_opt = torch.optim.Adam([v for v in dp.values() if v.requires_grad], lr=lr, betas=(0.0, 0.999), weight_decay=wd)
    with torch.no_grad():
        rand_labels = est_out(xr)
    xd, rl = x.to(DEV), rand_labels.to(DEV)
    for it in range(2):
        # ---- oracle iteration ----
        d_ref_opt.zero_grad(set_to_none=True)
        d_loss_ref, nb2 = O.update_discriminator_loss(gp, dp, est_out, x, rand_labels, None, False)
        for v in gp.values():
            v.grad = None
        d_loss_ref.backward()
        d_ref_opt.step()
        with torch.no_grad():
            for k, v in nb2.items():
                dp[k] = v.clone()
        g_ref_opt.zero_grad(set_to_none=True)
        g_ref = O.update_inference_loss(gp, dp, est_out, est_cpu, x, rand_labels, None, None, False, False)
        g_ref[0].backward()
        g_ref_opt.step()
        with torch.no_grad():                    # the power iteration of the G update's D forward (independent of the input)
            nb3 = O.sndisc_forward({k: v.detach() for k, v in dp.items()}, x, rand_labels, train=True)[1]
            for k, v in nb3.items():
                dp[k] = v.clone()
        # ---- build ----
        d_loss = st.update_discriminator(xd, rl, None)
        g_losses = st.update_inference(xd, rl, None, None)
        d_err = abs(d_loss.item() - d_loss_ref.item()) / max(1.0, abs(d_loss_ref.item()))
        g_err = abs(g_losses[0].item() - g_ref[0].item()) / max(1.0, abs(g_ref[0].item()))
        print(f"GAN iteration {it}: d_loss {d_loss.item():.6f} vs {d_loss_ref.item():.6f}, g_loss {g_losses[0].item():.6f} vs {g_ref[0].item():.6f}")
        assert d_err <= 3e-3 and g_err <= 3e-3, (it, d_err, g_err)
    # the weights themselves after two iterations: Adam with beta1 = 0 steps lr * g / (|g| + eps) per element -- where g ~ 0 the sign
    # differs between implementations, so single elements may be up to 2 * 2 * lr apart; the UPDATE as a whole must agree
    init = O.make_cunet_params(nc, seed)
    for k, prm in st.inference.named_parameters():
        if gp[k].grad is None:
            continue
        da = (prm.detach().cpu() - init[k]).reshape(-1).double()
        db = (gp[k].detach() - init[k]).reshape(-1).double()
        cos = (torch.dot(da, db) / (da.norm() * db.norm() + 1e-30)).item()
        assert cos >= 0.9 and (da - db).abs().mean().item() <= 0.2 * lr, (k, cos, (da - db).abs().mean().item())
