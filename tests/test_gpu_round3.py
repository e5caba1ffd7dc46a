"""Round 3 (VERDICT r2 "next" 1-3, 8, 9):
 * BASELINE configs[2] / configs[3] at their real sizes (GAN cls B=32, soft-label B=64, 256x256, bf16, ResNet-101 estimator in the
   loop): bitwise determinism of every G / D gradient and of D's power-iteration buffers, production bf16 kernels against the fp32
   generic kernels for SNDisc and the estimator at 256x256;
 * SURVEY 8f.1 as a real interchange test: a checkpoint written from the REFERENCE modules' state_dict(), and the sweeps the
   reference modules computed from it (tests/golden/ref_ckpt_*.npz);
 * evaluation() in the mode the reference runs it (D in train mode), against a fixture computed by the reference's own loop body;
 * a two-process data-parallel run of the REAL kernels (both ranks on cuda:0, gloo: RCCL refuses two ranks on one device);
 * the advisor's low-severity findings (1x1 head lane guard, l1_loss alignment).
"""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cunet_ref as O
from oracle import resnet_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD_TOL = {"fp32": 1e-3, "bf16": 5e-2}


def _cos(a, b):
    a, b = a.detach().double().reshape(-1).cpu(), b.detach().double().reshape(-1).cpu()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


# ---------------------------------------------------------------------------------------------------------------------------
# 8f.1: reference-written checkpoint -> build modules -> sweeps, against what the REFERENCE modules computed from the same file
# ---------------------------------------------------------------------------------------------------------------------------
def _unpack_ckpt(golden_dir, tmp_path, g):
    name = str(g["ckpt_file"])
    path = os.path.join(str(tmp_path), "run", name[:-3])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.open(os.path.join(golden_dir, name), "rb") as src, open(path, "wb") as dst:
        dst.write(src.read())
    return path


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_reference_checkpoint_interchange_and_sweeps(golden_dir, tmp_path, precision):
    """t_est_train.py:365-373 wrote the file (from the reference's own modules, tests/golden/make_golden.py:gen_checkpoint);
    ``load_checkpoint`` (weights_only=True) restores it into the HIP-backed modules; ``class_sweep`` (inf_transfer_c.py:114-121),
    ``signal_sweep`` (inf_transfer_e.py:136-143) and ``transfer_rows`` (inf_1year_signals.py:104) reproduce the outputs the
    REFERENCE modules computed from that file, within north_star's tolerances; the hipGraph path replays the same numbers;
    D's output likewise; and a checkpoint the build writes back is tensor-for-tensor the file it read."""
    import cunet
    import disc
    from wu.graph_infer import GraphedUNet
    from wu.infer_driver import (class_sweep, latest_checkpoint, load_checkpoint, normalize_minmax, save_checkpoint, signal_sweep,
                                 to_uint8, transfer_rows)
    g = np.load(os.path.join(golden_dir, "ref_ckpt_b2_64.npz"))
    batch, size, _, seed, nc, epoch, step = [int(v) for v in g["meta"]]
    path = _unpack_ckpt(golden_dir, tmp_path, g)
    G = cunet.Conditional_UNet(nc, precision=precision).to(DEV).eval()
    D = disc.SNDisc(nc, precision=precision).to(DEV).eval()
    assert load_checkpoint(latest_checkpoint(str(tmp_path), "run"), G, D) == (epoch, step)
    x, _ = O.make_inputs(batch, size, nc, seed, False)
    xd = x.to(DEV)
    tol = FWD_TOL[precision]
    sweep = class_sweep(G, xd)
    assert tuple(sweep.shape) == (nc, batch, 3, size, size)
    err_c = np.abs(sweep.cpu().numpy() - g["class_sweep"]).max()
    sig = signal_sweep(G, xd, torch.from_numpy(g["signals"]))
    err_s = np.abs(sig.cpu().numpy() - g["signal_sweep"]).max()
    rows = transfer_rows(G, xd, torch.from_numpy(g["per_row"]))
    err_r = np.abs(rows.cpu().numpy() - g["per_row_out"]).max()
    print(f"reference checkpoint, {precision}: class sweep max-abs {err_c:.3e}, signal sweep {err_s:.3e}, per-row {err_r:.3e} (tol {tol})")
    assert err_c <= tol and err_s <= tol and err_r <= tol
    # the sweep really is class-dependent at this size of error (class-to-class difference in the fixture: 0.2)
    assert np.abs(g["class_sweep"][0] - g["class_sweep"][1]).max() > 4 * tol or precision == "bf16"
    graphed = GraphedUNet(G, batch, size)
    assert torch.equal(class_sweep(G, xd, graphed=graphed), sweep)
    assert torch.equal(signal_sweep(G, xd, torch.from_numpy(g["signals"]), graphed=graphed), sig)
    # save_image(normalize=True) arithmetic on the GPU == the same formula on the CPU copy
    nm = normalize_minmax(sweep[0])
    cpu = sweep[0].cpu()
    for j in range(batch):
        lo, hi = float(cpu[j].min()), float(cpu[j].max())
        want = ((cpu[j] - lo) / (hi - lo + 1e-5)).clamp(0, 1)
        assert (nm[j].cpu() - want).abs().max().item() <= 1e-6
    assert to_uint8(nm).dtype == torch.uint8 and tuple(to_uint8(nm).shape) == (batch, size, size, 3)
    # D from the same file
    eye = torch.eye(nc)
    with torch.no_grad():
        d_out = D(torch.from_numpy(g["class_sweep"][1]).to(DEV), eye[1].expand(batch, nc).contiguous().to(DEV))[0]
    scale = max(1.0, float(np.abs(g["d_out"]).max()))
    assert np.abs(d_out.cpu().numpy() - g["d_out"]).max() <= (1e-3 if precision == "fp32" else 5e-2) * scale
    # and back: what the build writes is what it read (format and tensors)
    out_path = save_checkpoint(str(tmp_path), "back", G, D, epoch, step)
    a = torch.load(path, weights_only=True)
    b = torch.load(out_path, weights_only=True)
    assert set(a) == set(b) and a["epoch"] == b["epoch"] and a["global_step"] == b["global_step"]
    for part in ("inference", "discriminator"):
        assert list(a[part]) == list(b[part])                       # same keys in the same order
        for k in a[part]:
            assert a[part][k].dtype == b[part][k].dtype and torch.equal(a[part][k], b[part][k]), (part, k)


def test_axis_sweep_schedule():
    """demo.py:67-82: conditioning = estimator prediction with one axis replaced by alpha*sin(theta); against the oracle on the
    same rows (fp32)."""
    import cunet
    from wu.infer_driver import axis_sweep
    nc, seed, batch, size = 5, 12, 2, 32
    G = cunet.Conditional_UNet(nc, precision="fp32")
    gp = O.make_cunet_params(nc, seed)
    G.load_state_dict(gp)
    G = G.to(DEV).eval()
    x, pred = O.make_inputs(batch, size, nc, seed, True)
    thetas = [-np.pi / 2, 0.3]
    got = axis_sweep(G, x.to(DEV), pred.to(DEV), thetas, alpha=2.0)
    assert tuple(got.shape) == (2, nc, batch, 3, size, size)
    eye = torch.eye(nc)
    with torch.no_grad():
        for ti, th in enumerate(thetas):
            for a in (0, 3):
                c = torch.cat([eye[a:a + 1] * torch.sin(torch.tensor(th).float()) * 2.0] * batch) + torch.cat([1. - eye[a]] * batch).view(-1, nc) * pred
                want = O.cunet_forward(gp, x, c)
                assert (got[ti, a].cpu() - want).abs().max().item() <= 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# evaluation() with D in train mode: the reference's own loop body produced the fixture
# ---------------------------------------------------------------------------------------------------------------------------
class _PoolLinear(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b = torch.nn.Parameter(w, requires_grad=False), torch.nn.Parameter(b, requires_grad=False)

    def forward(self, t):
        return F.linear(F.adaptive_avg_pool2d(t, 8).flatten(1), self.w, self.b)


@pytest.mark.parametrize("max_images", [1024, 3])
def test_evaluation_train_mode_discriminator(golden_dir, max_images):
    """t_cls_train.py:331-352 with D in TRAIN mode (the reference never calls .eval()): the four means AND D's weight_u / weight_v
    after the sweep (2*B power iterations) against the fixture the reference modules produced (eval_dtrain_b3_32.npz)."""
    from wu.train_step import WeatherTransferStep
    g = np.load(os.path.join(golden_dir, "eval_dtrain_b3_32.npz"))
    bs, size, _, seed, nc = [int(v) for v in g["meta"]]
    est = _PoolLinear(torch.from_numpy(g["est_w"]), torch.from_numpy(g["est_b"]))
    st = WeatherTransferStep(nc, mode="cls", precision="fp32", device=DEV, ddp=False, seed=1, estimator=est)
    st.inference.load_state_dict(O.make_cunet_params(nc, seed))
    st.discriminator.load_state_dict(O.make_sndisc_params(nc, seed))
    st.inference.eval()
    st.discriminator.train()
    images, labels = O.make_inputs(bs, size, nc, seed, True)
    _, ref_labels = O.make_inputs(bs, size, nc, seed + 1, True)
    got, fake = st.evaluation(images.to(DEV), labels.to(DEV), ref_labels.to(DEV), max_images=max_images)
    assert tuple(fake.shape) == (bs, bs, 3, size, size)
    for k, v in zip([str(k) for k in g["mean_keys"]], g["means"]):
        assert abs(got[k].item() - float(v)) <= 2e-3 * max(1.0, abs(float(v))), (k, got[k].item(), float(v))
    sd = st.discriminator.state_dict()
    n = 0
    for k in sd:
        if k.endswith(("weight_u", "weight_v")):
            assert np.abs(sd[k].cpu().numpy() - g["buf_" + k]).max() <= 1e-4, k
            n += 1
    assert n == 20


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] / configs[3] at full size
# ---------------------------------------------------------------------------------------------------------------------------
def _full_step(mode, batch):
    from wu.resnet import ResNet101Estimator
    from wu.train_step import WeatherTransferStep
    nc = 5
    est = ResNet101Estimator(nc, precision="bf16")
    est.load_state_dict(R.make_resnet101_params(nc, 4), strict=False)
    st = WeatherTransferStep(nc, mode=mode, precision="bf16", device=DEV, ddp=False, seed=1, estimator=est)
    st.inference.load_state_dict(O.make_cunet_params(nc, 2))
    st.discriminator.load_state_dict(O.make_sndisc_params(nc, 2))
    for opt in (st.d_opt, st.g_opt):                 # lr 0: weights stay, so two iterations start from the same state
        for grp in opt.param_groups:
            grp["lr"] = 0.0
            grp["weight_decay"] = 0.0
    gen = torch.Generator().manual_seed(21)
    noise = [(torch.rand((batch, 3, 256, 256), generator=gen) * 2 - 1).to(DEV) for _ in range(2)]
    # With these weights D separates uniform noise from G's outputs by hundreds of hinge margins: d_loss would be exactly 0 and
    # every D gradient zero.  "Real" images are therefore drawn from G itself (other inputs / classes), and D's output bias is
    # centred on the 0.4 quantile of its real and fake outputs: roughly half the samples of each hinge branch are then active.
    eye = torch.eye(nc, device=DEV)
    with torch.no_grad():
        st.inference.eval()
        st.discriminator.eval()
        images = st.inference(noise[0], eye[torch.arange(batch) % nc]).clone()
        rand_images = st.inference(noise[1], eye[(torch.arange(batch) + 2) % nc]).clone()
        labels, pred = st.estimator(rand_images), st.estimator(images)
        st.discriminator.train()
        for _ in range(20):                          # converge the power iteration: the fill's u / v are random unit vectors, and
            st.discriminator(images[:2], pred[:2])   # u.(W v) of unconverged vectors is a tiny sigma (W / sigma ~ 1e2 per layer)
        st.discriminator.eval()
        real = st.discriminator(images, pred)[0]
        fake = st.discriminator(st.inference(images, labels), labels)[0]
        # (the 0.4 quantile, not the median: with equally many active samples in the two branches d d_loss / d l.bias is exactly 0)
        st.discriminator.l.bias -= torch.quantile(torch.cat([real, fake]).float().reshape(-1), 0.4)
        st.inference.train()
        st.discriminator.train()
    return st, images, rand_images


def _iteration(st, images, rand_images, d_state):
    st.discriminator.load_state_dict(d_state)        # the power-iteration buffers move 3 times per iteration: rewind them
    st.inference.dropout_seed = 5                    # Dropout ACTIVE (train mode), reproducible masks
    for p in list(st.inference.parameters()) + list(st.discriminator.parameters()):
        p.grad = None
    losses = st.step(images, rand_images)
    torch.cuda.synchronize()
    grads = {"G." + k: p.grad.clone() for k, p in st.inference.named_parameters() if p.grad is not None}
    grads.update({"D." + k: p.grad.clone() for k, p in st.discriminator.named_parameters() if p.grad is not None})
    bufs = {k: v.clone() for k, v in st.discriminator.state_dict().items() if k.endswith(("weight_u", "weight_v"))}
    return [float(v) for v in losses], grads, bufs


@pytest.mark.parametrize("mode,batch", [("cls", 32), ("est", 64)])
def test_full_size_gan_iteration_is_deterministic(mode, batch):
    """BASELINE configs[2] (t_cls_train GAN loop, B=32) and configs[3] (t_est_train soft-label loop, B=64 per GPU) at 256x256,
    bf16, Dropout active, the ResNet-101 estimator in the loop (t_cls_train.py:226-312,414-438 / t_est_train.py:214-283): two
    iterations from identical state are bitwise equal in all five losses, every G gradient (36), every D gradient (20) and D's
    20 power-iteration buffers; all finite.  Small shapes cannot show this: every persistent / prefetching kernel (LDS-DMA convs,
    stride-2 convs, the estimator's GEMMs, stem, pools) runs its multi-tile loops only at this size."""
    st, images, rand_images = _full_step(mode, batch)
    d_state = {k: v.clone() for k, v in st.discriminator.state_dict().items()}
    l1, g1, b1 = _iteration(st, images, rand_images, d_state)
    l2, g2, b2 = _iteration(st, images, rand_images, d_state)
    print(f"full-size GAN iteration {mode} B={batch}: losses {l1}")
    assert all(np.isfinite(v) for v in l1) and l1 == l2
    assert l1[0] > 0, "d_loss is exactly 0: the hinge is saturated and the D gradients would be trivially reproducible zeros"
    assert max(abs(v) for v in l1) < 1e4, "losses of an O(1)-scaled problem"
    assert len([k for k in g1 if k.startswith("G.")]) == 36 and len([k for k in g1 if k.startswith("D.")]) == 20
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        if k != "D.l.bias":                           # d d_loss / d l.bias = (#active fake - #active real) / B: 0 when all are active
            assert g1[k].abs().max().item() > 0, f"{k}: gradient is identically zero"
        assert torch.equal(g1[k], g2[k]), f"gradient of {k} is not reproducible at full size"
    assert len(b1) == 20
    moved = 0
    for k in b1:
        assert torch.equal(b1[k], b2[k]), f"buffer {k} is not reproducible"
        moved += int(not torch.equal(b1[k], d_state[k].to(DEV)))
    # the power iteration is ACTIVE in the step (three per iteration); the 1 x 512 head `l` is at its fixed point after one
    assert moved >= 12, f"only {moved} of 20 power-iteration buffers moved"
    # Round 4 (advisor): the overlapped schedule against the PLAIN order -- one more iteration with every stream overlap, the encoder
    # sharing and the fence-less ordering events switched off.  A missing wait or a stale operand that repeats run to run (the
    # cross-stream hazard of DESIGN.md 4 was 10 of 10) passes the two-run comparison above; it cannot pass this one.
    from wu import train_step as TS, unet_graph as UG
    # (also off in the plain order: the hipGraph replay of the estimator's no-grad pass, so the replay is compared with the eager launches)
    assert st._est_graphs and all(g.replays >= 2 for g in st._est_graphs.values()), "the estimator's no-grad pass did not run from its hipGraph"
    saved = (TS.OVERLAP_D_PASSES, TS.OVERLAP_D_WITH_ESTIMATOR, TS.SHARE_ENCODER, UG.LIGHT_EVENTS, TS.GRAPH_ESTIMATOR)
    try:
        TS.OVERLAP_D_PASSES = TS.OVERLAP_D_WITH_ESTIMATOR = TS.SHARE_ENCODER = TS.GRAPH_ESTIMATOR = False
        UG.LIGHT_EVENTS = False
        l3, g3, b3 = _iteration(st, images, rand_images, d_state)
    finally:
        TS.OVERLAP_D_PASSES, TS.OVERLAP_D_WITH_ESTIMATOR, TS.SHARE_ENCODER, UG.LIGHT_EVENTS, TS.GRAPH_ESTIMATOR = saved
    assert l1 == l3, f"losses differ between the overlapped and the plain order: {l1} vs {l3}"
    for k in g1:
        assert torch.equal(g1[k], g3[k]), f"gradient of {k}: overlapped order != plain order"
    for k in b1:
        assert torch.equal(b1[k], b3[k]), f"buffer {k}: overlapped order != plain order"


def test_full_size_sndisc_production_vs_generic():
    """SNDisc at 256x256 (disc.py:27-38): the bf16 production path (MFMA stride-1 LDS-DMA convs, stride-2 convs, the matrix-core
    first-layer data gradient) against the fp32 generic kernels on the same weights, B=4: the five outputs within 5e-2 of their
    scale, all 20 parameter gradients of the hinge loss and the input gradient aligned (cos >= 0.98 / 0.95)."""
    import disc
    nc, b = 5, 4
    p = O.make_sndisc_params(nc, 2)
    x, c = O.make_inputs(b, 256, nc, 3, True)
    res = {}
    for precision in ("fp32", "bf16"):
        D = disc.SNDisc(nc, precision=precision)
        D.load_state_dict(p)
        D = D.to(DEV).train()
        xd = x.to(DEV).requires_grad_(True)
        outs = D(xd, c.to(DEV))
        loss = torch.mean(torch.relu(1.0 - outs[0])) + torch.mean(torch.relu(1.0 + outs[0]))    # both hinge branches live
        loss.backward()
        res[precision] = ([o.detach().float() for o in outs], {k: q.grad.clone() for k, q in D.named_parameters()}, xd.grad.clone(),
                          {k: v.clone() for k, v in D.state_dict().items() if k.endswith(("weight_u", "weight_v"))})
    o32, g32, dx32, b32 = res["fp32"]
    o16, g16, dx16, b16 = res["bf16"]
    for i, (a, bb) in enumerate(zip(o16, o32)):
        scale = max(1.0, bb.abs().max().item())
        err = (a - bb).abs().max().item() / scale
        print(f"   SNDisc 256x256 output {i}: bf16 vs fp32 kernels err/scale {err:.3e} (scale {scale:.2f})")
        assert err <= 5e-2
    for k in b32:                                     # the power iteration is fp32 in both modes
        assert (b16[k] - b32[k]).abs().max().item() <= 1e-5, k
    assert len(g16) == 20
    for k in g16:
        cs = _cos(g16[k], g32[k])
        print(f"   SNDisc 256x256 grad {k:24s} cos {cs:.5f}")
        assert cs >= 0.98, f"{k}: cosine {cs}"
    assert _cos(dx16, dx32) >= 0.95


def test_full_size_estimator_production_vs_generic():
    """The frozen ResNet-101 (classifier.py:106-112) at 256x256, B=4: bf16 production kernels (1x1 GEMMs, MFMA 3x3, stem, pools)
    against the fp32 kernels on the same folded weights: outputs within 5e-2 of their scale, input-gradient cosine >= 0.975 (measured
    0.9847-0.988: the bf16 precision mode over 33 gated blocks -- block by block the HIP path agrees with a bf16 emulation at >= 0.9999,
    tests/test_gpu_round4.py::test_estimator_bf16_backward_block_by_block);
    B=32 forward run twice is bitwise identical and equals the B=4 slice rows to bf16 accuracy."""
    from wu.resnet import ResNet101Estimator
    nc = 5
    sd = R.make_resnet101_params(nc, 4)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand((32, 3, 256, 256), generator=gen) * 2 - 1)
    tgt = torch.rand((4, nc), generator=gen)
    res = {}
    for precision in ("fp32", "bf16"):
        est = ResNet101Estimator(nc, precision=precision)
        est.load_state_dict(sd, strict=False)
        est = est.to(DEV)
        xd = x[:4].to(DEV).requires_grad_(True)
        out = est(xd)
        F.mse_loss(out, tgt.to(DEV)).backward()
        res[precision] = (out.detach().clone(), xd.grad.clone(), est)
    scale = max(1.0, res["fp32"][0].abs().max().item())
    err = (res["bf16"][0] - res["fp32"][0]).abs().max().item() / scale
    cs = _cos(res["bf16"][1], res["fp32"][1])
    print(f"   ResNet-101 256x256 B=4: bf16 vs fp32 kernels out err/scale {err:.3e} (scale {scale:.2f}), input-grad cos {cs:.5f}")
    assert err <= 5e-2 and cs >= 0.975
    est = res["bf16"][2]
    with torch.no_grad():
        a = est(x.to(DEV))
        b = est(x.to(DEV))
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert (a[:4] - res["bf16"][0]).abs().max().item() <= 5e-2 * scale


# ---------------------------------------------------------------------------------------------------------------------------
# data parallel, two processes on the one GPU, REAL kernels
# ---------------------------------------------------------------------------------------------------------------------------
def test_two_process_data_parallel_real_kernels(tmp_path):
    """Two fresh ranks (``python -m torch.distributed.run --nproc-per-node 2``, both on cuda:0, backend gloo -- RCCL refuses two
    ranks on one device), each running the cUNet step with ``GradBucketReducer.attach`` (gradient sink + side stream) on its
    shard and then the GAN iteration with both reducers; rank 0 compares the averaged gradients with a single-process run on
    the concatenated batch and every rank's SN buffers after two iterations (tests/ddp_two_rank_worker.py).  The parent only
    waits for the child (no exec from a process that touched the GPU)."""
    out = os.path.join(str(tmp_path), "ddp.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WU_DDP_OUT=out, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    port = 29400 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_two_rank_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    print(r.stderr[-4000:])
    assert r.returncode == 0
    res = json.load(open(out))
    print(json.dumps(res, indent=1))
    assert res["world"] == 2
    assert res["unet_worst_rel"] <= 1e-4, res
    assert res["unet_buckets_launched_during_backward"] >= 1
    assert res["gan_g_worst_rel"] <= 1e-4 and res["gan_d_worst_rel"] <= 1e-4, res
    assert res["ranks_agree_bitwise"] and res["sn_buffers_agree_bitwise"] and res["params_agree_after_steps"]
    assert res["bf16_step_finite"] and res["bf16_ranks_agree_bitwise"]


# ---------------------------------------------------------------------------------------------------------------------------
# advisor (round 2), low severity
# ---------------------------------------------------------------------------------------------------------------------------
def test_conv1x1_tanh_lane_guard():
    """wu_conv1x1_tanh_fwd writes its 3 output channels from lanes 0..2 of a pixel group: Cin/E < 4 lanes per group must be
    REJECTED (it used to return rc 0 with channels >= Cin/E unwritten); the smallest legal widths compute the right thing."""
    from wu import kernels as K
    from wu.layout import as_nhwc, precision_code
    for p, cin, ok in (("bf16", 32, True), ("fp32", 16, True), ("bf16", 16, False), ("fp32", 8, False)):
        g = torch.Generator().manual_seed(cin)
        x = torch.rand((2, cin, 6, 10), generator=g) * 2 - 1
        if p == "bf16":
            x = x.to(torch.bfloat16).float()
        w = torch.rand((3, cin), generator=g) * 0.6 - 0.3
        b = torch.rand((3,), generator=g)
        xg = as_nhwc(x.to(DEV), precision_code(p))
        out = torch.full((2, 3, 6, 10), float("nan"), device=DEV)
        if ok:
            K.conv1x1_tanh(xg, w.to(DEV), b.to(DEV), out)
            ref = torch.tanh(F.conv2d(x, w.view(3, cin, 1, 1), b))
            assert (out.cpu() - ref).abs().max().item() <= 2e-5
        else:
            with pytest.raises(RuntimeError):
                K.conv1x1_tanh(xg, w.to(DEV), b.to(DEV), out)


def test_l1_loss_accepts_misaligned_views():
    """ops.l1_loss (reference ops.py:22-24 = F.l1_loss) on contiguous fp32 views at an odd storage offset: the fused kernel needs
    16-byte-aligned pointers, so these take the stock op instead of raising."""
    import ops
    g = torch.Generator().manual_seed(1)
    a = torch.rand(4099, generator=g).to(DEV)
    b = torch.rand(4099, generator=g).to(DEV)
    av = a[1:].requires_grad_(True)
    assert av.data_ptr() % 16 != 0
    loss = ops.l1_loss(av, b[1:])
    loss.backward()
    assert abs(loss.item() - F.l1_loss(a[1:].cpu(), b[1:].cpu()).item()) <= 1e-6
    assert av.grad is not None
    # aligned inputs still take the fused kernel (bitwise run-to-run)
    assert ops.l1_loss(a, b).item() == ops.l1_loss(a, b).item()


# ---------------------------------------------------------------------------------------------------------------------------
# bf16 gradients, stage by stage, with the upstream gradient held fixed (advisor r2 / VERDICT r2 weak #1)
# ---------------------------------------------------------------------------------------------------------------------------
def _f(t):
    return t.detach().float().cpu().contiguous()


@pytest.mark.parametrize("shape,train", [((1, 256, 256), True), ((2, 128, 96), False)])
def test_bf16_backward_stage_by_stage_vs_emulating_oracle(shape, train):
    """The end-to-end bf16 gradient comparison (test_gpu_round2.py::test_bf16_gradients_vs_emulating_oracle) is bounded by the
    instance-norm backward's amplification of one-ulp flips: cos 0.96-0.98 on the deep layers, in the emulation
    itself.  Here every stage of the REAL fused backward (wu/unet_graph.py; tensors taken from its tracing hook) is checked on
    its own: the oracle's bf16-emulating stage gets the SAME stage input and the SAME (bf16) upstream gradient the HIP stage
    had, so nothing is amplified and what is left is fp32 summation order inside one stage.  Required: every stage-input
    gradient and every one of the 36 parameter gradients cos >= 0.99999 (style layers 0.9999; measured 0.999999), at the model's own layer shapes
    (256x256: the LDS-DMA production kernels, gate-bit epilogues, fused pool / AdaIN-upsample-dropout kernels)."""
    from wu import functional as WF
    from wu import unet_graph
    import cunet
    n, h, w = shape
    nc, seed = 5, 14
    p0 = O.make_cunet_params(nc, seed)
    net = cunet.Conditional_UNet(nc, precision="bf16")
    net.load_state_dict(p0)
    net = net.to(DEV)
    net.train(train)
    net.dropout_seed = 33
    r = O._rng("stage", seed)
    x = torch.from_numpy(r.uniform(-1, 1, size=(n, 3, h, w)).astype(np.float32))
    c = torch.softmax(torch.from_numpy(r.standard_normal((n, nc)).astype(np.float32)), 1)
    masks = (None, None, None)
    if train:
        masks = [WF.dropout_mask(n, ch, h // d, w // d, 0.3, (33 * 4 + k) & 0x7FFFFFFFFFFFFFFF, torch.device(DEV)).float().cpu()
                 for k, ch, d in ((3, 512, 4), (2, 256, 2), (1, 128, 1))]
    cap = {}
    unet_graph.CAPTURE = cap
    try:
        xd = x.to(DEV)
        out = net(xd, c.to(DEV))
        torch.mean(torch.abs(out - xd)).backward()
        torch.cuda.synchronize()
    finally:
        unet_graph.CAPTURE = None
    T = {k: _f(v) for k, v in cap.items() if torch.is_tensor(v)}
    hip = {k: _f(q.grad) for k, q in net.named_parameters() if q.grad is not None}
    rows = []

    def check(what, got, want, lim):
        cs = _cos(got, want)
        rl = ((got.double() - want.double()).norm() / (want.double().norm() + 1e-30)).item()
        rows.append((what, cs, rl, lim))

    def params(*keys):
        return {k: p0[k].clone().requires_grad_(True) for k in keys}

    def block(name, xin_key, g_key, gx_key):
        """r_double_conv `name`: input T[xin_key], PRE-GATED upstream gradient T[g_key] -> dL/d(input) vs T[gx_key], dW, db."""
        keys = [f"{name}.0.weight", f"{name}.0.bias", f"{name}.2.weight", f"{name}.2.bias"]
        pp = params(*keys)
        xin = T[xin_key].clone().requires_grad_(gx_key is not None)
        src = O._q(xin, True) if name == "dconv_down1" else xin          # the image enters the first conv as a bf16 MFMA operand
        # T[g_key] is ALREADY gated by the sign of the HIP block's own output, so the emulation stops at the second conv's pre-activation:
        # gating once more by the sign of the EMULATED output drops the gradient of every element on which the two outputs (1e-3 apart)
        # disagree about > 0 -- one such element in 393 216 carrying 1e-5 of the gradient mass is cos 0.99999 by itself (round 4: measured
        # 0.999948 on dconv_up3 of the ragged case once the 16x12 layers ran on another kernel and b4 rounded differently, while the HIP
        # gradients agreed with an fp32 recomputation under their own gates to 1.000000; scratch/dbg_stage.py)
        mid = O._q(F.relu(O._conv3x3(src, O._qw(pp[keys[0]], True), pp[keys[1]], None)), True)
        y = O._conv3x3(mid, O._qw(pp[keys[2]], True), pp[keys[3]], None)
        gs = torch.autograd.grad(y, ([xin] if gx_key is not None else []) + [pp[k] for k in keys], T[g_key])
        if gx_key is not None:
            check(f"{name}: d input", T[gx_key], gs[0], 0.99999)
            gs = gs[1:]
        for k, g in zip(keys, gs):
            check(k, hip[k], g, 0.99999)

    def up_stage(lvl, xk, skipk, gk, gxk, mask):
        """cunet.py:59-62: AdaIN -> bilinear x2 -> dropout -> cat.  Input T[xk] (a ReLU output), upstream T[gk] (gradient of the
        concat buffer) -> gradient of the input GATED by ReLU'(input) vs T[gxk]; the style layer's dW / db."""
        name = f"adain{lvl}"
        keys = [f"{name}.l1.weight", f"{name}.l1.bias"]
        pp = params(*keys)
        xin = T[xk].clone().requires_grad_(True)
        a = O._RoundGradBF16.apply(O.adain(pp, name, xin, c))
        cat = O._q(torch.cat([O.dropout(O.upsample2(a), mask), T[skipk]], dim=1), True)
        gs = torch.autograd.grad(cat, [xin] + [pp[k] for k in keys], T[gk])
        check(f"{name}+upsample+dropout: d input (gated)", T[gxk], gs[0] * (T[xk] > 0).float(), 0.99999)
        for k, g in zip(keys, gs[1:]):
            check(k, hip[k], g, 0.9999)

    def pool_stage(catk, c0, gpk, gcatk, gxk):
        """cunet.py:46,49,52 backward + the skip gradient + the ReLU gate of the conv that produced the tensor."""
        conv = T[catk][:, c0:].clone().requires_grad_(True)
        pooled = F.max_pool2d(conv, 2)
        (gp,) = torch.autograd.grad(pooled, conv, T[gpk])
        want = (gp + T[gcatk][:, c0:]) * (conv.detach() > 0).float()
        check(f"maxpool bwd + skip ({catk})", T[gxk], want, 0.99999)

    # head (cunet.py:80-82): gradient of u1b gated by ReLU'(u1b); conv_last dW / db
    pp = params("conv_last.weight", "conv_last.bias")
    u1b = T["u1b"].clone().requires_grad_(True)
    o = torch.tanh(F.conv2d(u1b, pp["conv_last.weight"], pp["conv_last.bias"]))
    gs = torch.autograd.grad(o, [u1b, pp["conv_last.weight"], pp["conv_last.bias"]], T["gout"])
    check("head: d u1b (gated)", T["g_u1b"], gs[0] * (T["u1b"] > 0).float(), 0.99999)
    check("conv_last.weight", hip["conv_last.weight"], gs[1], 0.99999)
    check("conv_last.bias", hip["conv_last.bias"], gs[2], 0.99999)
    block("dconv_up1", "cat1", "g_u1b", "g_cat1")
    T["cat1_skip"], T["cat2_skip"], T["cat3_skip"] = T["cat1"][:, 128:], T["cat2"][:, 256:], T["cat3"][:, 512:]
    up_stage(1, "u2b", "cat1_skip", "g_cat1", "g_u2b", masks[2])
    block("dconv_up2", "cat2", "g_u2b", "g_cat2")
    up_stage(2, "u3b", "cat2_skip", "g_cat2", "g_u3b", masks[1])
    block("dconv_up3", "cat3", "g_u3b", "g_cat3")
    up_stage(3, "b4", "cat3_skip", "g_cat3", "g_b4", masks[0])
    block("dconv_down4", "p3", "g_b4", "g_p3")
    pool_stage("cat3", 512, "g_p3", "g_cat3", "g_conv3")
    block("dconv_down3", "p2", "g_conv3", "g_p2")
    pool_stage("cat2", 256, "g_p2", "g_cat2", "g_conv2")
    block("dconv_down2", "p1", "g_conv2", "g_p1")
    pool_stage("cat1", 128, "g_p1", "g_cat1", "g_conv1")
    block("dconv_down1", "x", "g_conv1", None)
    bad = []
    for what, cs, rl, lim in rows:
        print(f"   stage-wise bf16 {shape} {what:48s} cos {cs:.6f} rel {rl:.4f}")
        if not cs >= lim:
            bad.append(f"{what}: cos {cs:.6f} < {lim}")
    nparam = sum(1 for what, *_ in rows if what in hip)
    assert nparam == 36, nparam
    assert not bad, "; ".join(bad)


@pytest.mark.parametrize("size,batch", [(64, 2), (128, 2)])
def test_sndisc_bf16_gradients_vs_emulating_oracle(size, batch):
    """SNDisc in the bf16 production mode against ``O.sndisc_forward(emulate_bf16=True)`` (the reference's graph with the HIP
    path's storage points rounded to bf16, oracle/cunet_ref.py:sn_double_conv): outputs within 1e-2 of their scale, all 20
    parameter gradients of the hinge loss and the input gradient cos >= 0.999 -- no normalisation layer amplifies anything in D,
    so unlike G the end-to-end comparison is tight (the older checks in test_gpu_model.py accept err <= 0.5 rms)."""
    import disc
    nc, seed = 5, 3
    p = O.make_sndisc_params(nc, seed)
    x, c = O.make_inputs(batch, size, nc, seed, True)
    D = disc.SNDisc(nc, precision="bf16")
    D.load_state_dict(p)
    D = D.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    outs = D(xd, c.to(DEV))
    (torch.mean(torch.relu(1.0 - outs[0])) + torch.mean(torch.relu(1.0 + outs[0]))).backward()
    pg = {k: (v.clone().requires_grad_(True) if k.endswith(("weight_orig", "bias")) else v.clone()) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    outs_o, _ = O.sndisc_forward(pg, xr, c, train=True, emulate_bf16=True)
    (torch.mean(torch.relu(1.0 - outs_o[0])) + torch.mean(torch.relu(1.0 + outs_o[0]))).backward()
    for i, (a, b) in enumerate(zip(outs, outs_o)):
        scale = max(1.0, b.abs().max().item())
        err = (a.detach().float().cpu() - b.detach()).abs().max().item() / scale
        print(f"   SNDisc bf16 vs emulation, output {i}: err/scale {err:.3e}")
        assert err <= 1e-2
    worst = 1.0
    for k, q in D.named_parameters():
        cs = _cos(q.grad, pg[k].grad)
        print(f"   SNDisc bf16 vs emulation, grad {k:24s} cos {cs:.6f}")
        worst = min(worst, cs)
    cs_x = _cos(xd.grad, xr.grad)
    print(f"   SNDisc bf16 vs emulation, input gradient cos {cs_x:.6f}")
    assert worst >= 0.999 and cs_x >= 0.999


# ---------------------------------------------------------------------------------------------------------------------------------
# batched spectral normalisation (wu_spectral_norm_{fwd,bwd}_multi): the ten SN layers of SNDisc in one call
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("power_iter", [True, False])
def test_spectral_norm_multi_is_bitwise_the_single_weight_path(power_iter):
    """SNDisc's ten weight shapes (disc.py:11-24): W/sigma, the advanced u / v buffers and the weight gradients of the batched call
    equal ten single-weight calls bit for bit."""
    from wu import functional as WF
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    shapes = [(3, 3, 3, 3), (64, 3, 3, 3), (64, 64, 3, 3), (128, 64, 3, 3), (128, 128, 3, 3), (256, 128, 3, 3), (256, 256, 3, 3),
              (512, 256, 3, 3), (1, 512), (512, 5)]
    ws = [(torch.randn(s, generator=g) * 0.1).to(dev) for s in shapes]
    rows = [s[0] for s in shapes]
    cols = [int(np.prod(s[1:])) for s in shapes]
    us = [F.normalize(torch.randn(r, generator=g), dim=0).to(dev) for r in rows]
    vs = [F.normalize(torch.randn(c, generator=g), dim=0).to(dev) for c in cols]
    gs = [torch.randn(s, generator=g).to(dev) for s in shapes]

    def run(multi):
        w = [t.clone().requires_grad_(True) for t in ws]
        u, v = [t.clone() for t in us], [t.clone() for t in vs]
        if multi:
            out = WF.spectral_normalize_multi(w, u, v, power_iter, 1e-12)
        else:
            out = [WF.spectral_normalize(wi, ui, vi, power_iter, 1e-12) for wi, ui, vi in zip(w, u, v)]
        torch.autograd.backward(list(out), gs)
        return [o.detach() for o in out], u, v, [t.grad for t in w]

    a, b = run(False), run(True)
    for name, xs, ys in zip(("w_eff", "u", "v", "dw"), a, b):
        for i, (x, y) in enumerate(zip(xs, ys)):
            assert torch.equal(x, y), f"{name}[{i}] {shapes[i]} differs: max {(x - y).abs().max().item():.3e}"
    # and against torch's own arithmetic (fp32 round-off of a differently ordered dot product)
    for i in range(len(ws)):
        wm = ws[i].reshape(rows[i], -1)
        if power_iter:
            v_ref = F.normalize(wm.t() @ us[i], dim=0, eps=1e-12)
            u_ref = F.normalize(wm @ v_ref, dim=0, eps=1e-12)
        else:
            u_ref, v_ref = us[i], vs[i]
        sigma = torch.dot(u_ref, wm @ v_ref)
        assert torch.allclose(b[0][i], ws[i] / sigma, rtol=2e-5, atol=1e-7)
        assert torch.allclose(b[1][i], u_ref, rtol=1e-4, atol=1e-6) and torch.allclose(b[2][i], v_ref, rtol=1e-4, atol=1e-6)


def test_sndisc_batched_normalisation_matches_per_layer_path():
    """SNDisc.forward (one batched normalisation, SNLinear heads) against the same module with every layer normalising its own
    weight (the pre-batching path, forced by handing the layers no weights): outputs, advanced buffers and parameter gradients
    bit for bit, in training mode over two consecutive forwards (two power iterations) and in eval mode."""
    import disc
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    Da = disc.SNDisc(5, precision="bf16").to(dev)
    Db = disc.SNDisc(5, precision="bf16").to(dev)
    Db.load_state_dict(Da.state_dict())
    Db._normalize_weights = lambda: []
    g = torch.Generator(device="cpu").manual_seed(6)
    c = torch.softmax(torch.randn(4, 5, generator=g), dim=1).to(dev)
    for mode in ("train", "train", "eval"):
        x = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).to(dev)
        outs = []
        for D in (Da, Db):
            D.train(mode == "train")
            D.zero_grad(set_to_none=True)
            o = D(x, c)
            (o[0].sum() + o[4].float().mean()).backward()
            outs.append(o)
        for ta, tb in zip(outs[0], outs[1]):
            assert torch.equal(ta, tb)
        for (ka, ta), (kb, tb) in zip(Da.state_dict().items(), Db.state_dict().items()):
            assert ka == kb and torch.equal(ta, tb), ka
        for (ka, pa), (_, pb) in zip(Da.named_parameters(), Db.named_parameters()):
            assert pa.grad is not None and torch.equal(pa.grad, pb.grad), ka


def test_tiled_image_conv_beside_the_stem_mfma_kernel():
    """Regression test for the cross-stream hazard of DESIGN.md 4: the LDS-tiled image-layout 3 -> 3 conv (forward and its
    transposed-weight data-gradient form) must give bit-identical results when it runs on a second stream BESIDE the stem's MFMA
    kernel (the arrangement of WeatherTransferStep.update_inference: D's first conv beside the estimator's stem).  With packed-FP32
    FMAs in the kernel 10 of 10 launches came back with a few wrong 16-lane groups (round 4's discriminating run, profiles/r04_hazard.txt:
    every v_pk_fma_f32 form fails whatever its LDS-read queue, every scalar-FMA form is exact); a blocker on the main stream makes the two
    kernels start together.  Round 4: the weight-gradient form (img3_wgrad_kernel) is covered too."""
    from wu import _lib, kernels as K, resnet as RN
    from wu.layout import empty_nhwc
    from wu.unet_graph import _side_stream
    dev = torch.device("cuda:0")
    B, S, code = 32, 256, _lib.BF16
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
    w33 = ((torch.rand((3, 3, 3, 3), generator=g) - 0.5) * 0.3).to(dev)
    b3 = (torch.rand(3, generator=g) - 0.5).to(dev)
    ws = ((torch.rand((64, 3, 7, 7), generator=g) - 0.5) * 0.1).to(dev)
    bs = (torch.rand(64, generator=g) - 0.5).to(dev)
    stem_y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    gy3 = (torch.rand((B, 3, S, S), generator=g) - 0.5).to(dev)

    def wgrad(out):
        # img3_wgrad_kernel (81 + 3 accumulators per thread over a grid-stride loop of tiles): dW in out[:81], dbias in out[81:84]
        K.conv3x3_c3_wgrad(x, gy3, out[:81].view(3, 3, 3, 3), out[81:84], 1, code, dy_nchw=True)

    subjects = {"forward": (lambda out: K.conv3x3_c3(x, w33, b3, out, 1, 0, True, code), lambda: torch.full_like(x, float("nan"))),
                "data gradient": (lambda out: K.conv3x3_c3_dgrad(x, w33, out, 1, code, dy_nchw=True), lambda: torch.full_like(x, float("nan"))),
                "weight gradient": (wgrad, lambda: torch.full((84,), float("nan"), device=dev))}
    RN.stem7x7(x, ws, bs, stem_y, 1, code)
    torch.cuda.synchronize()
    stem_ref = stem_y.clone()
    for name, (subj, fresh) in subjects.items():
        ref = fresh()
        subj(ref)
        torch.cuda.synchronize()
        assert not ref.isnan().any()
        for _ in range(6):
            out = fresh()
            torch.cuda.synchronize()
            torch.cuda._sleep(3_000_000)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                subj(out)
            RN.stem7x7(x, ws, bs, stem_y, 1, code)
            main.wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"{name}: differs beside the stem kernel (max {(out - ref).abs().nan_to_num(1e9).max().item():.3e})"
            assert torch.equal(stem_y, stem_ref)


# ---------------------------------------------------------------------------------------------------------------------------------
# encoder sharing between the two generator forwards of one GAN iteration (wu/train_step.py, Conditional_UNet.forward encoder_cache)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_encoder_cache_is_bitwise_neutral():
    """G(x, c1) under no_grad followed by G(x, c2) with grad -- the reference's two generator forwards of an iteration
    (t_cls_train.py:302,242), Dropout active with different masks -- give the same outputs and the same 36 parameter gradients
    bit for bit whether the second forward recomputes the encoder or reuses the first one's (cunet.py:45-54 has no Dropout and does
    not see the condition)."""
    import cunet
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.rand((3, 3, 64, 96), generator=g) * 2 - 1).to(dev)
    c1 = torch.softmax(torch.randn(3, 5, generator=g), dim=1).to(dev)
    c2 = torch.softmax(torch.randn(3, 5, generator=g), dim=1).to(dev)
    gout = torch.randn(3, 3, 64, 96, generator=g).to(dev)
    torch.manual_seed(8)
    net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()

    def run(shared):
        net.dropout_seed = 11
        net.zero_grad(set_to_none=True)
        cache = {} if shared else None
        with torch.no_grad():
            o1 = net(x, c1, cache)
        net.dropout_seed = 12
        o2 = net(x, c2, cache)
        o2.backward(gout)
        if shared:
            assert cache.get("computed") == 1 and cache.get("reused") == 1
        return o1.clone(), o2.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(a[2]) == 36 and a[2].keys() == b[2].keys()
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    # another input must not hit the cache
    cache = {}
    with torch.no_grad():
        net(x, c1, cache)
        net(x.clone(), c1, cache)
    assert cache["computed"] == 2 and "reused" not in cache


def test_gan_iteration_with_and_without_encoder_sharing():
    """A whole GAN iteration (stand-in estimator, every stream overlap on) from identical state with the encoder shared and with
    it recomputed: the five losses, every G and D gradient and D's power-iteration buffers bit for bit."""
    from wu import train_step as TS
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(31)
    x = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).to(dev)
    xr = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).to(dev)
    st = TS.WeatherTransferStep(5, mode="cls", precision="bf16", device=dev, ddp=False, seed=2)
    for opt in (st.d_opt, st.g_opt):
        for grp in opt.param_groups:
            grp["lr"] = 0.0
            grp["weight_decay"] = 0.0
    d_state = {k: v.clone() for k, v in st.discriminator.state_dict().items()}
    res = []
    try:
        for share in (True, False, True):
            TS.SHARE_ENCODER = share
            res.append(_iteration(st, x, xr, d_state))
    finally:
        TS.SHARE_ENCODER = True
    for other in res[1:]:
        assert res[0][0] == other[0]
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], other[1][k]), k
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], other[2][k]), k
