"""hipGraph-captured inference (BASELINE.json configs[4]: 512x512 inf_transfer_c-style forward).

The inference drivers of the reference call ``transfer(batch, cond)`` in a loop with static shapes
(inference/inf_transfer_c.py:114-121, inf_transfer_e.py:136-143, demo.py:75-81).  One forward is ~40 kernel
launches plus the AdaIN style kernels; captured once into a hipGraph, a replay is a single launch from the host.
Every launcher of libwu_kernels.so is capture-safe: no allocation, no synchronisation, no host reads
(include/wu_kernels.h).

Dropout.  Only demo.py:54 calls ``transfer.eval()``; inf_transfer_c.py:88-96, inf_transfer_e.py:98-107 and the eval
scripts run the generator in TRAIN mode, i.e. with Dropout(0.3) ACTIVE.  Captured kernel arguments are frozen, so a naive
capture would replay one mask for ever; here the dropout kernels add a DEVICE-RESIDENT counter to their captured seed
(``seed_dev`` of wu_adain_upcat_fwd) and the graph itself bumps that counter at the end of every replay: replay k of a
train-mode graph draws the masks the eager module would draw with ``dropout_seed = base_seed + k``.

Weights.  The captured kernels read the packed MFMA operand tensors that existed at capture time.  The graph keeps those
tensors alive, records every parameter's version, and re-captures when a version has moved (optimizer step,
load_state_dict) instead of replaying stale -- or, after the allocator recycled them, garbage -- weights.
"""
import torch

from .unet_graph import BLOCKS


class GraphedUNet:
    """Static-shape ``Conditional_UNet`` forward captured in a hipGraph (eval mode, or train mode = dropout active).

        g = GraphedUNet(net, batch=16, size=512)
        out = g(x, c)          # copies into the static inputs, replays, returns the static output tensor
    """

    def __init__(self, net, batch, size, num_classes=None, height=None, warmup=2, base_seed=None):
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedUNet needs the module on a GPU")
        if not net.fused:
            raise ValueError("GraphedUNet captures the fused single-node schedule (net.fused = True)")
        nc = num_classes if num_classes is not None else net.adain1.num_classes
        h = height if height is not None else size
        self.net = net
        self.training = bool(net.training)
        self.warmup = warmup
        self.x = torch.zeros((batch, 3, h, size), dtype=torch.float32, device=dev)
        self.c = torch.zeros((batch, nc), dtype=torch.float32, device=dev)
        # dropout-active graphs: seeds = (base_seed * 4 + k) + counter, counter += 4 per replay  (cunet._next_seed)
        self.base_seed = int(base_seed if base_seed is not None else (net.dropout_seed if net.dropout_seed is not None
                                                                       else torch.initial_seed() & 0xFFFFFFF))
        self.seed_counter = torch.zeros(1, dtype=torch.int64, device=dev) if self.training else None
        self.replays = 0
        self._capture()

    def _params(self):
        return list(self.net.parameters())

    def _capture(self):
        net, dev = self.net, self.x.device
        if bool(net.training) != self.training:
            raise RuntimeError("GraphedUNet: the module's train/eval mode changed since construction")
        saved = (net.dropout_seed, net._seed_dev)
        net.dropout_seed, net._seed_dev = (self.base_seed, self.seed_counter) if self.training else saved
        try:
            self.graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):   # first-use work (weight packing, attribute setup) must not be captured
                    net(self.x, self.c)
            torch.cuda.current_stream(dev).wait_stream(side)
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = net(self.x, self.c)
                if self.seed_counter is not None:
                    self.seed_counter.add_(4)                      # next replay: the masks of dropout_seed + 1
        finally:
            net.dropout_seed, net._seed_dev = saved
        # the captured kernels hold raw pointers into these: keep them alive for the graph's lifetime
        self._held = []
        for name in BLOCKS:
            blk = getattr(net, name)
            for conv in (blk[0], blk[2]):
                self._held.append((conv._packed.w_fwd, conv._packed.w_dgrad))
        # fused optimizers update parameters without moving ``_version`` (wu.functional): the optimizer-step generation is part of
        # the freshness test
        from .functional import _WEIGHT_GENERATION
        self._versions = (_WEIGHT_GENERATION[0],) + tuple(p._version for p in self._params())
        self._ptrs = tuple(p.data_ptr() for p in self._params())

    def _check_fresh(self):
        ps = self._params()
        from .functional import _WEIGHT_GENERATION
        if (_WEIGHT_GENERATION[0],) + tuple(p._version for p in ps) != self._versions or tuple(p.data_ptr() for p in ps) != self._ptrs:
            self._capture()            # weights changed since capture: re-pack + re-capture (never replay stale operands)

    def __call__(self, x, c, copy_out=False):
        self.x.copy_(x, non_blocking=True)
        self.c.copy_(c, non_blocking=True)
        return self.replay(copy_out)

    def replay(self, copy_out=False):
        """Replay on the inputs already resident in ``self.x`` / ``self.c``."""
        self._check_fresh()
        self.graph.replay()
        self.replays += 1
        return self.out.clone() if copy_out else self.out

    def set_seed_offset(self, k):
        """Dropout-active graphs: make the NEXT replay draw the masks of ``dropout_seed = base_seed + k``."""
        if self.seed_counter is None:
            raise RuntimeError("set_seed_offset: this graph was captured in eval mode (no dropout)")
        self.seed_counter.fill_(4 * int(k))
