"""Data-parallel host logic on CPU: world_size-2 gloo processes (no GPU, no HIP compute).  Checks that the
bucketed reducer (a) averages gradients exactly, (b) launches buckets in reverse-parameter order as they
fill during backward, (c) tolerates parameters that never receive a gradient (AdaIN.emb, utils.py:32),
(d) gives the same result as a single process on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _toy(seed):
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 30), torch.nn.Tanh(), torch.nn.Linear(30, 1))
    m.unused = torch.nn.Parameter(torch.ones(7))       # never used in forward
    return m


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _toy(seed=rank)                            # replicas start DIFFERENT: broadcast must fix that
        red = GradBucketReducer(list(m.parameters()), bucket_mb=1e-3)     # ~262 floats per bucket -> several buckets
        torch.manual_seed(123)
        x, y = torch.randn(8, 6), torch.randn(8, 1)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        for _ in range(2):                             # two steps: zero_grad must reset state
            red.zero_grad()
            loss = torch.mean((m(xs) - ys) ** 2)
            loss.backward()
            log_before_finalize = list(red.launch_log)
            red.finalize()
        # numpy arrays are pickled by value (tensors would travel as shared-memory fds that die with this process)
        grads = [p.grad.detach().numpy().copy() for p in m.parameters()]
        w0 = [p.detach().numpy().copy() for p in m.parameters()]
        q.put((rank, grads, w0, log_before_finalize, list(red.launch_log), len(red.buckets)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, w0, log0, full0, nb), (_, g1, w1, log1, full1, _) = res
    # identical replicas (rank-0 weights) and identical averaged gradients
    import numpy as np
    for a, b in zip(w0, w1):
        assert np.array_equal(a, b)
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)
    g0 = [torch.from_numpy(g) for g in g0]
    # single-process reference on the full batch with rank 0's weights
    m = _toy(seed=0)
    torch.manual_seed(123)
    x, y = torch.randn(8, 6), torch.randn(8, 1)
    torch.mean((m(x) - y) ** 2).backward()
    for p, g in zip(m.parameters(), g0):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(g, ref, atol=1e-6), (g - ref).abs().max()
    # buckets launched during backward, in order 0,1,2,.. (reverse parameter order = readiness order); the bucket
    # holding the unused parameter is only launched by finalize()
    assert nb >= 3 and log0 == sorted(log0) and len(log0) >= nb - 1
    assert sorted(full0) == list(range(nb)) and log0 == log1


class _SinkLinear(torch.autograd.Function):
    """Stand-in for the fused cUNet node: a two-layer MLP as ONE autograd node whose backward, when a reducer is attached,
    accumulates each layer's weight gradient straight into the bucket views and announces it (wu.unet_graph.UNetFn does the
    same with its HIP weight-gradient kernels) -- and returns None for those parameters."""

    @staticmethod
    def forward(ctx, sink, x, w1, w2):
        h = torch.tanh(x @ w1.t())
        ctx.save_for_backward(x, w1, w2, h)
        ctx.sink, ctx.P = sink, (w1, w2)
        return h @ w2.t()

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, h = ctx.saved_tensors
        sink, (p1, p2) = ctx.sink, ctx.P
        order = []
        dw2 = g.t() @ h
        gh = (g @ w2) * (1 - h * h)
        dw1 = gh.t() @ x
        if sink is None:
            return None, None, dw1, dw2
        p2.grad.add_(dw2); sink.grad_written(p2)        # layer 2 is ready first
        p1.grad.add_(dw1); sink.grad_written(p1)
        return None, None, None, None


def _sink_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        w1 = torch.nn.Parameter(torch.randn(300, 6))
        w2 = torch.nn.Parameter(torch.randn(2, 300))
        tail = torch.nn.Linear(2, 1)                   # gradients through ordinary autograd hooks, ready BEFORE the node's
        # ready order: tail (autograd reaches it first), then w2, w1; tiny buckets -> one parameter each
        red = GradBucketReducer(list(tail.parameters()) + [w2, w1], bucket_mb=1e-4, ready_order=True)
        torch.manual_seed(123)
        x, y = torch.randn(8, 6), torch.randn(8, 1)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        for _ in range(2):
            red.zero_grad()
            out = tail(_SinkLinear.apply(red, xs, w1, w2))
            torch.mean((out - ys) ** 2).backward()
            log = list(red.launch_log)
            red.finalize()
        q.put((rank, [p.grad.detach().numpy().copy() for p in (w1, w2, tail.weight, tail.bias)], log, len(red.buckets)))
    finally:
        dist.destroy_process_group()


def test_direct_write_sink_world2():
    """The fused-node path: gradients written into bucket views from inside one autograd node, buckets launched in ready
    order while that node is still running, result == single-process gradient of the full batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    (_, g0, log0, nb), (_, g1, log1, _) = res
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)
    torch.manual_seed(5)
    w1 = torch.nn.Parameter(torch.randn(300, 6))
    w2 = torch.nn.Parameter(torch.randn(2, 300))
    tail = torch.nn.Linear(2, 1)
    torch.manual_seed(123)
    x, y = torch.randn(8, 6), torch.randn(8, 1)
    torch.mean((tail(_SinkLinear.apply(None, x, w1, w2)) - y) ** 2).backward()
    for g, p in zip(g0, (w1, w2, tail.weight, tail.bias)):
        assert torch.allclose(torch.from_numpy(g), p.grad, atol=1e-5), (torch.from_numpy(g) - p.grad).abs().max()
    assert nb == 3 and log0 == [0, 1, 2] and log1 == log0           # every bucket launched during backward, in ready order


def test_ready_order_covers_all_parameters():
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    import cunet
    from wu.ddp import ready_order
    net = cunet.Conditional_UNet(5)
    order = ready_order(net)
    assert {id(p) for p in order} == {id(p) for p in net.parameters() if p.requires_grad} and len(order) == len(set(map(id, order)))
    assert order[0] is net.conv_last.weight and order[2] is net.dconv_up1[2].weight


def test_reducer_single_process_noop():
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer
    m = _toy(0)
    red = GradBucketReducer(list(m.parameters()), bucket_mb=1.0)
    red.zero_grad()
    torch.mean(m(torch.ones(2, 6))).backward()
    red.finalize()
    assert all(p.grad is not None for p in m.parameters())
    with pytest.raises(ValueError):
        GradBucketReducer([])


# ---------------------------------------------------------------------------------------------------------------------------
# several backward passes per optimizer step (micro-batches): a gradient that reaches a bucket whose all-reduce is already in
# flight must raise; inside `with reducer.accumulate():` the launches wait for finalize() and the sums are exact
# ---------------------------------------------------------------------------------------------------------------------------
def _accum_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _toy(seed=3)
        red = GradBucketReducer(list(m.parameters()), bucket_mb=1e-3)
        torch.manual_seed(77)
        x, y = torch.randn(16, 6), torch.randn(16, 1)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        halves = [(xs[:4], ys[:4]), (xs[4:], ys[4:])]
        # (a) unguarded second backward: must raise, not silently race
        red.zero_grad()
        torch.mean((m(halves[0][0]) - halves[0][1]) ** 2).backward()
        raised = False
        try:
            torch.mean((m(halves[1][0]) - halves[1][1]) ** 2).backward()
        except RuntimeError as e:
            raised = "accumulate" in str(e)
        red.finalize()
        # (b) the supported way
        red.zero_grad()
        with red.accumulate():
            for xa, ya in halves:
                (0.5 * torch.mean((m(xa) - ya) ** 2)).backward()
            launched_inside = list(red.launch_log)
        red.finalize()
        q.put((rank, raised, launched_inside, [p.grad.detach().numpy().copy() for p in m.parameters()]))
    finally:
        dist.destroy_process_group()


def test_micro_batch_accumulation_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for _, raised, launched_inside, _ in res:
        assert raised, "a gradient arriving after its bucket's all-reduce was launched must raise"
        assert launched_inside == [], "accumulate(): nothing may be launched before finalize()"
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)
    m = _toy(seed=3)
    torch.manual_seed(77)
    x, y = torch.randn(16, 6), torch.randn(16, 1)
    torch.mean((m(x) - y) ** 2).backward()          # == mean over ranks of (0.5 * mse(half 1) + 0.5 * mse(half 2))
    for g, p in zip(res[0][3], m.parameters()):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(torch.from_numpy(g), ref, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------------
# the fused node's REAL gradient routing (wu.unet_graph.GradRouter) with two producer streams, on recording fake streams:
# every announcement is made on the side stream after it waited for the main stream, so a bucket's collective (ordered after
# the stream that is current at the announcement) can never start before a producer on either stream
# ---------------------------------------------------------------------------------------------------------------------------
class _FakeStream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_stream(self, other):
        self.log.append(("wait", self.name, other.name))


class _FakeOps:
    def __init__(self, main):
        self.cur = [main]

    def current(self):
        return self.cur[-1]

    def order_after(self, waiter, producer):
        waiter.wait_stream(producer)

    def use(self, stream):
        ops = self

        class _Ctx:
            def __enter__(self_):
                ops.cur.append(stream)

            def __exit__(self_, *exc):
                ops.cur.pop()
                return False
        return _Ctx()


def _router_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, shard_batch
    from wu.unet_graph import GradRouter
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(11)
        # three "layers" (weight, bias): parameter list laid out like UNetFn's P (iw, iw + 1 pairs)
        P = [torch.nn.Parameter(torch.randn(*shp)) for shp in ((40, 6), (40,), (30, 40), (30,), (1, 30), (1,))]
        order = [P[4], P[5], P[2], P[3], P[0], P[1]]                     # ready order: last layer first
        red = GradBucketReducer(order, bucket_mb=1e-4, ready_order=True)  # one parameter per bucket
        torch.manual_seed(123)
        x, y = torch.randn(8, 6), torch.randn(8, 1)
        xs, ys = shard_batch(x, rank, world), shard_batch(y, rank, world)
        log = []
        main, side = _FakeStream("main", log), _FakeStream("side", log)
        ops = _FakeOps(main)
        announce_stream = []
        orig = red.grad_written

        def spy(p):
            announce_stream.append(ops.current().name)
            log.append(("announce", ops.current().name))
            orig(p)
        red.grad_written = spy
        red.zero_grad()
        # forward + manual backward of a 3-layer MLP, gradients routed exactly as UNetFn.backward routes them
        h1 = torch.tanh(xs @ P[0].t() + P[1]); h2 = torch.tanh(h1 @ P[2].t() + P[3]); out = h2 @ P[4].t() + P[5]
        g = 2 * (out - ys) / out.numel()
        router = GradRouter(red, P, [tuple(p.shape) for p in P], lambda shp: torch.empty(shp), ops, main, side)
        with torch.no_grad():
            # layer 3: a "thin" layer whose gradient kernel runs on the MAIN stream (like conv_last / the first conv)
            dw, db, acc = router.bufs(4)
            dw.copy_(g.t() @ h2); db.copy_(g.sum(0)); router.done(("l3", 0), 4, dw, db)
            g2 = (g @ P[4]) * (1 - h2 * h2)
            # layers 2, 1: weight-gradient kernels on the SIDE stream
            for key, iw, gin, inp in (("l2", 2, g2, h1), ("l1", 0, None, xs)):
                if gin is None:
                    gin = (g2 @ P[2]) * (1 - h1 * h1)
                dw, db, acc = router.bufs(iw)
                router.on_side(lambda: (dw.copy_(gin.t() @ inp), db.copy_(gin.sum(0)), router.done((key, 0), iw, dw, db)))
        launched = list(red.launch_log)
        red.finalize()
        q.put((rank, [p.grad.detach().numpy().copy() for p in P], launched, announce_stream, log, len(red.buckets)))
    finally:
        dist.destroy_process_group()


def test_fused_node_grad_router_two_streams_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_router_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    (_, g0, launched, ann, log, nb), (_, g1, launched1, ann1, _, _) = res
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)
    assert nb == 6 and launched == list(range(6)) == launched1        # every bucket launched from inside "backward", in ready order
    assert ann == ["side"] * 6 == ann1                                 # announcements always on the side stream ...
    for i, ev in enumerate(log):                                       # ... each preceded by side.wait_stream(main)
        if ev[0] == "announce":
            before = [e for e in log[:i] if e[0] == "wait"]
            assert before and before[-1] == ("wait", "side", "main")
    # == single-process gradient of the full batch
    torch.manual_seed(11)
    P = [torch.nn.Parameter(torch.randn(*shp)) for shp in ((40, 6), (40,), (30, 40), (30,), (1, 30), (1,))]
    torch.manual_seed(123)
    x, y = torch.randn(8, 6), torch.randn(8, 1)
    h1 = torch.tanh(x @ P[0].t() + P[1]); h2 = torch.tanh(h1 @ P[2].t() + P[3])
    torch.mean((h2 @ P[4].t() + P[5] - y) ** 2).backward()
    for g, p in zip(g0, P):
        assert torch.allclose(torch.from_numpy(g), p.grad, atol=1e-5), (torch.from_numpy(g) - p.grad).abs().max()


# ---------------------------------------------------------------------------------------------------------------------------
# a collective moves no ``_version``: every derived-state cache (packed MFMA operands, SN identity, captured graphs, the folded
# estimator) must be invalidated by the broadcast helpers themselves (advisor finding, round 2)
# ---------------------------------------------------------------------------------------------------------------------------
def _bcast_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, broadcast_buffers
    from wu import functional as WF
    import nets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)
        conv = nets.Conv3x3(64, 64)
        sn = nets.SNConv3x3(64, 64)
        code = 1
        # what PackedConv.get() records after packing (the pack itself is a HIP launch; the key logic is host code)
        conv._packed.key = conv._packed.make_key(conv.weight, code)
        sn._packed.key = sn._packed.make_key(sn.weight_orig, code, sn.weight_ident())
        v0 = (conv.weight._version, sn.weight_u._version)
        ext0 = WF._EXTERNAL_GENERATION[0]
        red = GradBucketReducer(list(conv.parameters()), broadcast=False)
        fresh_before = not conv._packed.stale(conv.weight, code)
        red.broadcast_parameters()                                       # a LATE broadcast: after the "forward" that packed
        stale_w = conv._packed.stale(conv.weight, code)
        sn._packed.key = sn._packed.make_key(sn.weight_orig, code, sn.weight_ident())
        broadcast_buffers(sn)
        stale_sn = sn._packed.stale(sn.weight_orig, code, sn.weight_ident())
        v1 = (conv.weight._version, sn.weight_u._version)
        q.put((rank, fresh_before, stale_w, stale_sn, v0 == v1, WF._EXTERNAL_GENERATION[0] - ext0,
               conv.weight.detach().numpy().copy(), sn.weight_u.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_broadcast_invalidates_packed_operands_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for _, fresh_before, stale_w, stale_sn, version_unmoved, ext_bumps, _, _ in res:
        assert fresh_before
        assert stale_w, "broadcast_parameters() must invalidate the packed conv operands"
        assert stale_sn, "broadcast_buffers() must invalidate the spectral-norm identity"
        assert ext_bumps == 2
        # the premise: if torch ever starts bumping _version on collectives this still passes; it documents why we bump
        assert version_unmoved in (True, False)
    assert np.array_equal(res[0][6], res[1][6]) and np.array_equal(res[0][7], res[1][7])


# ---------------------------------------------------------------------------------------------------------------------------------------------
# tail_mb (round 4): the LAST bucket's collective is the one nothing overlaps (profiles/r04_fake_collective.txt), so the last-ready parameters
# get a small bucket of their own.  Layout on the real generator / discriminator, and a world-2 gloo run that the split changes nothing but
# the launch schedule.
# ---------------------------------------------------------------------------------------------------------------------------------------------
def test_tail_bucket_layout_on_the_real_networks():
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    import cunet
    import disc
    from wu.ddp import GradBucketReducer, ready_order
    net = cunet.Conditional_UNet(5, precision="fp32")
    order = ready_order(net)
    plain = GradBucketReducer(order, bucket_mb=12.0, ready_order=True)
    sizes_plain = [b["flat"].numel() * 4 for b in plain.buckets]
    plain.remove_hooks()
    red = GradBucketReducer(order, bucket_mb=12.0, ready_order=True, tail_mb=1.0)
    sizes = [b["flat"].numel() * 4 for b in red.buckets]
    assert len(sizes) == len(sizes_plain) + 1 and sizes[:-2] == sizes_plain[:-1] and sizes[-2] + sizes[-1] == sizes_plain[-1]
    assert sizes[-1] <= (1 << 20) and sizes[-1] + red.buckets[-2]["params"][-1].numel() * 4 > (1 << 20)     # the LARGEST suffix that fits
    # same parameters, same order, every .grad a view of its bucket
    flat_order = [p for b in red.buckets for p in b["params"]]
    assert len(flat_order) == len(order) and all(a is b for a, b in zip(flat_order, order))
    for b in red.buckets:
        for p in b["params"]:
            assert p.grad.untyped_storage().data_ptr() == b["flat"].untyped_storage().data_ptr()
    # the tail holds the first encoder block and everything autograd delivers after the fused node (AdaIN style layers)
    tail = {id(p) for p in red.buckets[-1]["params"]}
    assert all(id(p) in tail for p in net.dconv_down1.parameters()) and all(id(p) in tail for p in net.adain1.parameters())
    assert not any(id(p) in tail for p in net.dconv_down4.parameters())
    red.remove_hooks()
    # discriminator (registration order reversed = readiness order): one 9.3-MB bucket becomes a big one + a tail of the large-image layers
    d = disc.SNDisc(5, precision="fp32")
    dr = GradBucketReducer(list(d.parameters()), bucket_mb=12.0, tail_mb=1.0)
    ds = [b["flat"].numel() * 4 for b in dr.buckets]
    assert len(ds) == 2 and ds[1] <= (1 << 20) < ds[0]
    tail = {id(p) for p in dr.buckets[-1]["params"]}
    assert all(id(p) in tail for p in d.conv1.parameters()) and not any(id(p) in tail for p in d.conv4.parameters())
    dr.remove_hooks()
    # one parameter per bucket already: nothing to split; a tail limit smaller than the last parameter: nothing split either
    m = _toy(0)
    assert len(GradBucketReducer(list(m.parameters()), bucket_mb=1e-6, tail_mb=1.0).buckets) == len(list(m.parameters()))
    assert len(GradBucketReducer(list(m.parameters()), bucket_mb=1.0, tail_mb=1e-9).buckets) == 1


def _tail_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
    from wu.ddp import GradBucketReducer, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for tail in (None, 200 * 4 / (1 << 20)):            # tail limit = 200 floats
            m = _toy(seed=0)
            red = GradBucketReducer(list(m.parameters()), bucket_mb=1.0, tail_mb=tail)       # one bucket without the split
            torch.manual_seed(123)
            x, y = torch.randn(8, 6), torch.randn(8, 1)
            red.zero_grad()
            torch.mean((m(shard_batch(x, rank, world)) - shard_batch(y, rank, world)) ** 2).backward()
            inside = list(red.launch_log)
            red.finalize()
            out.append(([p.grad.detach().numpy().copy() for p in m.parameters()], inside, [len(b["params"]) for b in red.buckets]))
            red.remove_hooks()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_tail_bucket_world2_same_gradients_earlier_launch():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_tail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ((g_plain, in_plain, lay_plain), (g_tail, in_tail, lay_tail)) in res:
        assert lay_plain == [7] and len(lay_tail) == 2 and sum(lay_tail) == 7, (lay_plain, lay_tail)
        for a, b in zip(g_plain, g_tail):
            assert np.array_equal(a, b)                   # an average of the same two numbers either way
        # the single bucket holds the unused parameter: only finalize() launches it; with the split the first-ready part goes out during backward
        assert in_plain == [] and in_tail == [0], (in_plain, in_tail)
    for a, b in zip(res[0][1][1][0], res[1][1][1][0]):
        assert np.array_equal(a, b)
