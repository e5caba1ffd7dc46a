"""Data parallelism for the cUNet / SNDisc training steps: one process per GPU, gradients averaged with
bucketed all-reduce over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm), launched from
autograd hooks so the collectives overlap the remaining backward conv kernels.

The reference is single-GPU (t_cls_train.py:38-39); this is new functionality.  The path
shards naturally: every sample is independent in G (AdaIN is per-sample instance norm; no BatchNorm on
the executed path) and in D, and every loss is a batch mean over equal-size shards, so the mean of the
per-rank gradients equals the global-batch gradient.

Design for xGMI (point-to-point links, ring collectives are per-link bound): the payload is small
(31.2 MB fp32 for G, 9.3 MB for D) next to a multi-TFLOP step, so the goal is overlap and few launches,
not bandwidth: a few buckets, filled in REVERSE parameter order (decoder gradients are ready first), each
bucket one flat fp32 buffer that the parameters' ``.grad`` tensors are views of (no gather/scatter copies).
Round 4 (stand-in collective kernel beside the persistent conv kernels, profiles/r04_fake_collective.txt): the LAST
bucket's collective is exposed whole and every overlapped one costs about a third of its duration, so the shipped
layout is one big bucket that completes well before the end of the backward plus a small tail (``tail_mb``).
"""
import torch
import torch.distributed as dist


def _invalidate_packed():
    from .functional import invalidate_packed      # late: wu.functional loads the HIP library's ctypes table
    invalidate_packed()


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradBucketReducer:
    """Bucketed, overlapped gradient averaging.

    ``params``     parameters in registration (forward) order; buckets are formed over the reversed list
                   (``ready_order=True``: the list already IS the order in which gradients become ready)
    ``bucket_mb``  target bucket size; the last-registered (first-ready) parameters fill bucket 0
    ``tail_mb``    if given: the last-ready parameters (largest suffix of the last bucket within this size, at least one parameter stays)
                   become a bucket of their own -- the one collective nothing can overlap is then a small one

    Gradients reach a bucket either through autograd (post-accumulate hooks) or, for the fused cUNet node whose
    gradients would otherwise all surface at once when the node returns, directly: ``attach(net)`` makes the node write
    each layer's dW / db into the bucket views the moment its weight-gradient kernel is enqueued and call
    ``grad_written`` -- a full bucket's all-reduce then runs beside the REST of the backward pass.

    Usage per step:  ``reducer.zero_grad(); loss.backward(); reducer.finalize(); optimizer.step()``.
    Several backward passes per optimizer step (micro-batch accumulation, a network used twice in one graph):
    ``with reducer.accumulate(): ...backward passes...`` defers every collective to ``finalize()``; without it a gradient
    that reaches a bucket whose all-reduce has already been launched raises (it would race with the collective and
    never be reduced).
    ``param.grad`` is a view into its bucket's flat buffer for the lifetime of the reducer (do not call
    ``optimizer.zero_grad(set_to_none=True)``; use ``reducer.zero_grad()``).
    """

    def __init__(self, params, bucket_mb=12.0, group=None, broadcast=True, ready_order=False, tail_mb=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucketReducer: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("GradBucketReducer: parameters must be fp32 on one device")
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []          # list of dict(flat, params, pending, work)
        groups, cur, cur_n = [], [], 0
        for p in (self.params if ready_order else reversed(self.params)):
            if cur and cur_n + p.numel() > limit:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        # The LAST bucket completes with the last gradient of the backward pass: nothing is left to run beside its collective, its whole
        # duration is added to the step (measured with a stand-in collective kernel: profiles/r04_fake_collective.txt).  ``tail_mb`` splits
        # the last-ready parameters (the largest suffix that fits) into a bucket of their own, so that the exposed collective is a
        # latency-sized one and the bytes before it travel beside the remaining kernels.
        if tail_mb is not None and len(groups[-1]) > 1:
            tail_limit = int(tail_mb * (1 << 20) / 4)
            last, k, n = groups[-1], len(groups[-1]), 0
            while k > 1 and n + last[k - 1].numel() <= tail_limit:
                k -= 1
                n += last[k].numel()
            if k < len(last):
                groups[-1:] = [last[:k], last[k:]]
        for grp in groups:
            self.buckets.append(self._make_bucket(grp, dev))
        self._bucket_of = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._bucket_of[p] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self._avg_op = backend == "nccl"       # RCCL has a native AVG; gloo needs SUM + scale
        self._written = set()                  # parameters whose gradient was written since zero_grad()
        self._direct = set()                   # ... of those, the ones announced through grad_written() (direct bucket writes)
        self.launch_log = []                   # bucket indices in launch order (tests / tracing)
        self.enabled = True                    # False: gradients deposited by a backward are ignored (not reduced)
        self.defer = False                     # True (accumulate()): no launch before finalize()
        if broadcast and self.world > 1:
            self.broadcast_parameters()

    @staticmethod
    def _make_bucket(params, dev):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return {"flat": flat, "params": list(params), "pending": len(params), "work": None, "launched": False}

    def broadcast_parameters(self, src=0):
        """Replicas start identical (weights and, for SNDisc, the power-iteration buffers via the module's
        own broadcast_buffers call).  A collective does NOT move ``p._version`` (measured: 0 -> 0 across dist.broadcast and
        dist.all_reduce, Parameter or plain tensor), which the packed MFMA operand cache (wu.functional.PackedConv), the SN
        identity (nets.SNConv3x3.weight_ident), captured graphs (wu.graph_infer) and the folded estimator (wu.resnet) key on:
        the process-wide generation is bumped instead, so a broadcast AFTER a forward (re-sync, resume on rank 0 then
        broadcast, a reducer built late) cannot leave a rank convolving with stale operands."""
        with torch.no_grad():
            for p in self.params:
                dist.broadcast(p, src=src, group=self.group)
        _invalidate_packed()

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["work"] = None
            b["launched"] = False
        self.launch_log.clear()
        self._written.clear()
        self._direct.clear()

    def fresh(self, p):
        """True until a gradient of `p` has been written since zero_grad(): the first writer may OVERWRITE the (zeroed) bucket
        view instead of read-modify-writing it; later backward passes without zero_grad() must accumulate."""
        return p not in self._written

    def _launch(self, bi):
        b = self.buckets[bi]
        if b["launched"]:
            return
        b["launched"] = True
        self.launch_log.append(bi)
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
            # async: the process group's own stream waits for the gradients enqueued so far on the compute
            # stream, then the collective runs beside the rest of backward
            b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def _on_grad_ready(self, p, direct=False):
        if direct:
            self._direct.add(p)
        elif p in self._direct:
            # autograd runs a parameter's post-accumulate hooks even when the node returned None for it (torch >= 2.x): for a
            # parameter whose gradient the fused node already wrote and announced itself this is not a second gradient
            return
        self._written.add(p)               # any arrival (autograd hook or direct write): later writers must accumulate
        if not self.enabled:
            return
        bi = self._bucket_of[p]
        b = self.buckets[bi]
        if p.grad.data_ptr() < b["flat"].data_ptr() or p.grad.data_ptr() >= b["flat"].data_ptr() + b["flat"].numel() * 4:
            raise RuntimeError("GradBucketReducer: a parameter's .grad was replaced (use reducer.zero_grad(), "
                               "not zero_grad(set_to_none=True))")
        if b["launched"] and self.world > 1:
            raise RuntimeError("GradBucketReducer: a gradient arrived for a bucket whose all-reduce is already in flight "
                               "(second backward / micro-batch without zero_grad()): wrap the backward passes in "
                               "`with reducer.accumulate():` so the collectives wait for finalize()")
        b["pending"] -= 1
        if b["pending"] == 0 and not self.defer:
            self._launch(bi)

    def grad_written(self, p):
        """A producer accumulated this step's gradient of `p` straight into ``p.grad`` (the bucket view) on the current
        stream and returns None for it to autograd: same bookkeeping as the autograd hook."""
        self._on_grad_ready(p, direct=True)

    def accumulate(self):
        """Context manager: gradients of several backward passes are summed in the buckets, nothing is launched until
        ``finalize()`` (called after the block)."""
        red = self

        class _Defer:
            def __enter__(self_):
                red.defer = True
                return red

            def __exit__(self_, *exc):
                red.defer = False
                return False
        return _Defer()

    def owns(self, params):
        return all(p in self._bucket_of for p in params)

    def attach(self, net):
        """Route the fused cUNet node's weight gradients through ``grad_written`` (see class docstring)."""
        net.grad_sink = self
        return self

    def finalize(self):
        """Launch the buckets that never filled (parameters unused in this backward, e.g. AdaIN.emb,
        utils.py:32), wait for every collective, and finish the mean."""
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                if not self._avg_op:
                    b["flat"].div_(self.world)
                b["work"] = None

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()


def ready_order(net):
    """Parameters of a Conditional_UNet in the order their gradients become ready in the fused backward: head and decoder
    convs first, then the encoder back to front; everything autograd produces AFTER the node returns (AdaIN style layers,
    cunet.py:56-58 / utils.py:41-46) last, so only one small bucket is left to reduce after the last conv kernel."""
    convs = []
    for name in ("conv_last", "dconv_up1", "dconv_up2", "dconv_up3", "dconv_down4", "dconv_down3", "dconv_down2", "dconv_down1"):
        mod = getattr(net, name)
        if name == "conv_last":
            convs.extend(mod.parameters())
        else:        # backward visits the second conv of a block before the first
            convs.extend(list(mod[2].parameters()) + list(mod[0].parameters()))
    seen = {id(p) for p in convs}
    rest = [p for p in net.parameters() if id(p) not in seen]
    return [p for p in convs + rest if p.requires_grad]


def broadcast_buffers(module, src=0, group=None):
    """Keep SN ``weight_u`` / ``weight_v`` (nets.py:28-31) identical across ranks."""
    if not is_distributed():
        return
    with torch.no_grad():
        for b in module.buffers():
            dist.broadcast(b, src=src, group=group)
    _invalidate_packed()     # collectives do not move ``_version``: bump the generation every derived-state cache keys on


def shard_batch(batch, rank, world):
    """Contiguous equal shards of the global minibatch (global_batch % world == 0)."""
    n = batch.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return batch[rank * per:(rank + 1) * per]
