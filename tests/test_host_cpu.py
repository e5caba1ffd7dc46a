"""CPU: host logic that needs no GPU -- the ops.py losses / helpers against their reference definitions, the keep-mask
packing used for injected dropout masks, the content-hash build stamps, bench.py's self-launch of N ranks (gloo rendezvous
only), the oracle's bf16-emulation mode (sanity: it is the fp32 graph with rounding points), and the batched evaluation()
identity the product relies on."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cunet_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ops_losses_match_reference_definitions():
    import ops
    torch.manual_seed(0)
    fake, real = torch.randn(6, 1), torch.randn(6, 1)
    assert torch.equal(ops.dis_hinge(fake, real), torch.mean(torch.relu(1. - real)) + torch.mean(torch.relu(1. + fake)))   # ops.py:42-45
    assert torch.equal(ops.gen_hinge(fake), torch.mean(-fake))                                                             # ops.py:47-48
    a, b = torch.randn(4, 3, 8, 8), torch.randn(4, 3, 8, 8)
    assert torch.equal(ops.l1_loss(a, b), F.l1_loss(a, b)) and torch.equal(ops.adv_loss(a, b), F.mse_loss(a, b))
    with pytest.raises(AssertionError):
        ops.l1_loss(a, b[:2])
    logits, soft, idx = torch.randn(5, 5), torch.softmax(torch.randn(5, 5), 1), torch.tensor([0, 3, 2, 4, 1])
    assert torch.equal(ops.pred_loss(logits, soft), torch.nn.MSELoss()(logits, soft))                                      # ops.py:37-39
    assert torch.equal(ops.pred_loss(logits, idx, one_hot=True), torch.nn.CrossEntropyLoss()(logits, idx))                 # ops.py:30-36
    maps_a, maps_b = [torch.randn(2, 4, 4, 4), torch.randn(2, 8, 2, 2)], [torch.randn(2, 4, 4, 4), torch.randn(2, 8, 2, 2)]
    assert torch.allclose(ops.feat_loss(maps_a, maps_b), torch.mean(torch.stack([F.l1_loss(u, v) for u, v in zip(maps_a, maps_b)])))
    v = torch.tensor([[0.1, 0.9], [0.7, 0.2], [0.2, 0.3]])
    ref = torch.zeros_like(v).scatter_(0, torch.argmax(v, 0, keepdim=True), 1)                                            # ops.py:50-54
    assert torch.equal(ops.vector_to_one_hot(v), ref)
    assert torch.equal(ops.make_table_img(a, None, [b, a]), torch.cat([a, b, a], dim=2))
    # `from ops import *` must leak the names the scripts rely on (t_cls_train.py:328 uses F)
    ns = {}
    exec("from ops import *", ns)
    for name in ("F", "Variable", "np", "nn", "torch", "gen_hinge", "dis_hinge", "l1_loss", "adv_loss", "pred_loss",
                 "get_rand_labels", "get_sequential_labels", "Variable_Float", "make_table_img", "soft_transform"):
        assert name in ns, name


def test_pack_keep_mask_layout():
    from wu.kernels import pack_keep_mask
    g = torch.Generator().manual_seed(5)
    for dtype, e in ((torch.bfloat16, 8), (torch.float32, 4)):
        m = (torch.rand((2, 16, 3, 5), generator=g) < 0.7).to(torch.uint8)
        bits = pack_keep_mask(m, dtype).numpy()
        n, c, h, w = m.shape
        assert bits.shape == (n * h * w * c // e,)
        for (ni, hi, wi, ch) in ((0, 0, 0, 0), (1, 2, 4, 1), (0, 1, 3, c // e - 1)):
            byte = bits[((ni * h + hi) * w + wi) * (c // e) + ch]
            for k in range(e):
                assert (byte >> k) & 1 == int(m[ni, ch * e + k, hi, wi])


def test_build_stamp_is_content_hash(tmp_path, monkeypatch):
    from wu import _build
    if _build.is_stale():
        _build.build(verbose=False)
    assert not _build.is_stale()
    h0 = _build.source_hash()
    # a changed source byte -> stale, regardless of mtimes
    real = _build.sources
    fake_src = tmp_path / "extra.hip"
    fake_src.write_text("// new kernel\n")
    monkeypatch.setattr(_build, "sources", lambda: real() + [str(fake_src)])
    assert _build.source_hash() != h0 and _build.is_stale()


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks (child processes through the stock launcher)
    and rank 0 prints ONE JSON line; --launch-check keeps it to the rendezvous (gloo), so it runs without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": 2, "n_gpus": 2}


def test_oracle_bf16_emulation_is_the_same_graph_with_rounding_points():
    nc, seed = 5, 2
    p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
    x, c = O.make_inputs(1, 32, nc, seed, True)
    y32 = O.cunet_forward(p, x, c)
    y16, st = O.cunet_forward(p, x, c, emulate_bf16=True, return_stages=True)
    assert 0 < (y32 - y16).abs().max().item() <= 5e-2            # a bf16-sized perturbation, not a different function
    for k in ("conv1", "conv2", "conv3", "bottleneck", "up3", "up2", "up1"):
        t = st[k].detach()
        assert torch.equal(t, t.to(torch.bfloat16).float()), k    # every parked tensor is exactly representable in bf16
    O.bench_loss(y16, x).backward()
    g = p["dconv_down2.0.weight"].grad
    assert g is not None and g.dtype == torch.float32 and not torch.equal(g, g.to(torch.bfloat16).float())   # dW stays fp32


def test_batched_evaluation_identity():
    """evaluation() (t_cls_train.py:314-367) averages per-pass means over B passes; one pass over the B*B (image, condition)
    pairs gives the same four numbers because every pass has the same size -- the identity the product's batched
    evaluation() (wu/train_step.py) relies on.  Checked on the oracle restatement itself."""
    nc, seed, bs = 5, 4, 3
    gp, dp = O.make_cunet_params(nc, seed), O.make_sndisc_params(nc, seed)
    images, labels = O.make_inputs(bs, 32, nc, seed, True)
    _, ref_labels = O.make_inputs(bs, 32, nc, seed + 1, True)
    lin = torch.nn.Linear(3 * 32 * 32, nc)
    est = lambda t: lin(t.flatten(1))
    looped, fakes = O.evaluation(gp, dp, est, est, images, labels, ref_labels)
    with torch.no_grad():
        cond = ref_labels.repeat_interleave(bs, dim=0)
        xx = images.repeat(bs, 1, 1, 1)
        fake = O.cunet_forward(gp, xx, cond)
        fd = O.sndisc_forward(dp, fake, cond, train=False)[0][0]
        rd = O.sndisc_forward(dp, images, labels, train=False)[0][0]
        batched = {"g_loss_adv": O.gen_hinge(fd).item(), "g_loss_l1": F.l1_loss(fake, xx).item(),
                   "g_loss_w": F.mse_loss(est(fake), cond).item(), "d_loss": O.dis_hinge(fd, rd).item()}
    assert torch.allclose(fake.view(bs, bs, 3, 32, 32), torch.stack(fakes), atol=1e-5)
    for k in looped:
        assert abs(looped[k] - batched[k]) <= 1e-5 * max(1.0, abs(looped[k])), (k, looped[k], batched[k])


def test_resnet_estimator_state_dict_keys_are_torchvisions():
    """wu.resnet.ResNet101Estimator keeps torchvision's module names, so a torchvision resnet101 state-dict loads unchanged
    (classifier.py:106-112 / estimator.py:143-151 build that model); checked against the oracle's restatement of the key list."""
    from oracle import resnet_ref as R
    from wu.resnet import ResNet101Estimator
    net = ResNet101Estimator(5)
    have = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    want = R.resnet101_param_shapes(5)
    assert have == want and len(want) == 522                     # torchvision resnet101: 626 entries incl. 104 num_batches_tracked
    # the one published number the restatement can be held to: torchvision's resnet101 has 44,549,160 parameters (1000 classes)
    assert sum(int(np.prod(s)) for k, s in R.resnet101_param_shapes(1000).items() if not k.endswith(("running_mean", "running_var"))) == 44_549_160
    assert not any(p.requires_grad for p in net.parameters()) and not net.training
    net.train()
    assert not net.training                                      # frozen for good (t_cls_train.py:173,178)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 64, 64))


def test_packed_weight_cache_follows_optimizer_steps():
    """torch.optim.Adam(fused=True) updates parameters without moving ``_version`` (measured on the GPU box: 1 -> 1), so the packed
    operand cache keys on an optimizer-step generation as well: any optimizer.step() makes every PackedConv stale, and so does
    invalidate_packed(); an in-place update through autograd-visible ops is caught by ``_version`` as before."""
    import torch
    from wu import functional as WF
    w = torch.nn.Parameter(torch.ones(64, 64, 3, 3))
    pc = WF.PackedConv()
    pc.key = pc.make_key(w, 1)                       # as if packed now
    assert not pc.stale(w, 1)
    w.grad = torch.ones_like(w)
    torch.optim.SGD([w], lr=0.1).step()
    assert pc.stale(w, 1)
    pc.key = pc.make_key(w, 1)
    # an optimizer that touches OTHER parameters only: still stale (the generation is process-wide; a spare repack is cheap)
    other = torch.nn.Parameter(torch.ones(3)); other.grad = torch.ones(3)
    torch.optim.Adam([other], lr=0.1).step()
    assert pc.stale(w, 1)
    pc.key = pc.make_key(w, 1)
    WF.invalidate_packed()
    assert pc.stale(w, 1)
    pc.key = pc.make_key(w, 1)
    with torch.no_grad():
        w.add_(1.0)
    assert pc.stale(w, 1)


def test_bf16_sum_order_witness():
    """The checked basis of the bf16 gradient tolerances (tests/test_gpu_round2.py::test_bf16_gradients_vs_emulating_oracle
    accepts the HIP gradient within 2x the emulation-to-emulation distance).  Two CPU runs of the SAME bf16-emulating graph that
    differ only in the order their convs visit the input channels (``sum_order``: the same real-number function, another fp32
    rounding sequence):
      * in fp32 the re-ordering is invisible (every gradient cos >= 0.9999999);
      * in the bf16 emulation it is not -- the layers downstream of the last AdaIN (dconv_up1.*, conv_last.*, adain1.l1.*) still
        agree to >= 0.999, every gradient upstream of an instance-norm backward only to 0.95-0.996 (measured B=2 64x64: up3.2.bias
        0.951, the encoder 0.975-0.995).
    So an end-to-end bf16 gradient cosine of 0.95-0.98 on the deep layers is the precision mode, not a kernel property -- which
    is why the kernels' own arithmetic is pinned stage by stage with the upstream gradient held fixed (test_gpu_round3.py)."""
    nc, seed, n, size = 5, 13, 2, 64
    r = O._rng("emu", seed)
    x = torch.from_numpy(r.uniform(-1, 1, size=(n, 3, size, size)).astype(np.float32))
    c = torch.softmax(torch.from_numpy(r.standard_normal((n, nc)).astype(np.float32)), 1)
    grads = {}
    for tag, emu, order in (("emu", True, None), ("emu2", True, 1), ("fp32", False, None), ("fp32b", False, 1)):
        p = {k: v.clone().requires_grad_(True) for k, v in O.make_cunet_params(nc, seed).items()}
        O.bench_loss(O.cunet_forward(p, x, c, None, emulate_bf16=emu, sum_order=order), x).backward()
        grads[tag] = {k: v.grad for k, v in p.items() if v.grad is not None}

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return (torch.dot(a, b) / (a.norm() * b.norm())).item()

    assert len(grads["emu"]) == 36
    deep = []
    for k in grads["emu"]:
        assert cos(grads["fp32"][k], grads["fp32b"][k]) >= 0.9999999, k
        ee = cos(grads["emu"][k], grads["emu2"][k])
        if k.startswith(("dconv_up1", "conv_last", "adain1")):
            assert ee >= 0.999, (k, ee)
        else:
            assert 0.9 <= ee, (k, ee)
            deep.append(ee)
    assert len(deep) == 28 and min(deep) <= 0.98 and max(deep) <= 0.9975      # the amplification is there, on every deep layer


def test_dropout_hash_statistics():
    """The dropout keep decisions (cunet.py:61,68,75: nn.Dropout(0.3)) come from a counter hash of (seed, element group): the splitmix64
    finaliser, 16 bits per element.  Pinned on the numpy restatement (tests/_dropout_hash.py; the GPU test pins the kernels to it bit for
    bit): keep rate of each 16-bit field, its histogram, independence of the four fields of a draw, of neighbouring groups (lag 1 and one
    384-channel pixel) and of ADJACENT seeds (the three dropout layers of a step use seeds 4 s + 1, 2, 3).  The cheaper candidate measured
    in round 4 (rand4_cheap: as good statistically, not faster on the GPU, not shipped) goes through the same checks."""
    import numpy as np
    from _dropout_hash import rand4, rand4_cheap, keep_thr
    for rand4 in (rand4, rand4_cheap):
        _check_dropout_hash(rand4, keep_thr)


def _check_dropout_hash(rand4, keep_thr):
    import numpy as np
    n = 1 << 22
    g = np.arange(n, dtype=np.uint64)
    thr = keep_thr(0.3)
    assert thr == 45875
    sig_rate, sig_corr = (0.21 / n) ** 0.5, n ** -0.5

    def fields(r):
        return [((r >> np.uint64(16 * e)) & np.uint64(0xFFFF)).astype(np.int64) for e in range(4)]

    keeps = {}
    for seed in (0, 133, 134, 135, (1 << 40) + 7):
        f = fields(rand4(seed, g))
        k = [(x < thr).astype(np.float64) for x in f]
        keeps[seed] = k
        for e in range(4):
            assert abs(k[e].mean() - thr / 65536.0) < 5 * sig_rate, (seed, e, k[e].mean())
            h = np.bincount(f[e] >> 6, minlength=1024)
            chi2 = float(((h - n / 1024) ** 2 / (n / 1024)).sum())
            assert abs(chi2 - 1023) < 6 * (2 * 1023) ** 0.5, (seed, e, chi2)
        for i in range(4):
            for j in range(4):
                if j < i:
                    assert abs(np.corrcoef(k[i], k[j])[0, 1]) < 5 * sig_corr
                assert abs(np.corrcoef(k[i][:-1], k[j][1:])[0, 1]) < 5 * sig_corr          # the next group
                assert abs(np.corrcoef(k[i][:-96], k[j][96:])[0, 1]) < 5 * sig_corr        # the same channels of the next pixel (C = 384)
    for a, b in ((133, 134), (134, 135), (0, 133)):
        for e in range(4):
            assert abs(np.corrcoef(keeps[a][e], keeps[b][e])[0, 1]) < 5 * sig_corr, (a, b, e)
    # every output bit is fair
    r = rand4(7, g)
    for bit in range(64):
        assert abs(float(((r >> np.uint64(bit)) & np.uint64(1)).mean()) - 0.5) < 5 * 0.5 * sig_corr, bit
