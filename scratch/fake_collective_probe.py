#!/usr/bin/env python3
"""One-GPU stand-in for the question a multi-rank run would answer: what happens to the training step when a collective's
kernel needs CUs while every CU is owned by a persistent 160-KiB-LDS conv workgroup.  RCCL with one rank launches nothing (profiles/r04_rccl_overlap.txt),
so each bucket's all-reduce is replaced by a STAND-IN kernel on a communication stream (scratch/micro/fake_coll.hip: G workgroups x 256 threads that hold
LDS -- so they cannot share a CU with a conv workgroup -- and stay resident for T microseconds), launched exactly where GradBucketReducer launches the
collective (bucket complete, from inside the fused backward) and awaited in finalize().  bench.py's step otherwise (B=32 256x256 bf16, fused Adam).

    python scratch/fake_collective_probe.py        # on the GPU box; prints one line per variant and round
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "weather-unet_amd")]
import torch
import cunet
import ops
from wu import _lib, ddp
from wu.ddp import GradBucketReducer, ready_order
from wu import unet_graph as UG

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_SO = os.path.join(ROOT, "scratch", "micro", "fake_coll.so")
if not os.path.exists(_SO):            # the .so is git-ignored: build it where the script runs (hipcc is on the GPU box too)
    import subprocess
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", _SO[:-3] + ".hip", "-o", _SO])
FC = ctypes.CDLL(_SO)
FC.fake_coll_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
comm = torch.cuda.Stream(dev)
CFG = {"grid": 0, "usec": 0, "lds": 16384}
LOG = []


class FakeWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream(dev).wait_event(self.ev)


def fake_all_reduce(t, op=None, group=None, async_op=False):
    cur = torch.cuda.current_stream(dev)
    comm.wait_stream(cur)                                  # the collective orders after the gradients enqueued so far (what ProcessGroupNCCL does)
    with torch.cuda.stream(comm):
        if CFG["grid"] > 0:
            # usec < 0: duration from the message size -- 25 us + bytes / 70 GB/s (an 8-rank ring all-reduce over xGMI is per-link bound; assumption, see header)
            usec = CFG["usec"] if CFG["usec"] >= 0 else int(25 + t.numel() * 4 / (CFG.get("gbs", 70.0) * 1e3))
            rc = FC.fake_coll_launch(CFG["grid"], CFG["lds"], usec, ctypes.c_void_p(comm.cuda_stream))
            assert rc == 0, rc
        ev = torch.cuda.Event()
        ev.record(comm)
    LOG.append(t.numel())
    return FakeWork(ev)


torch.manual_seed(0)
net = cunet.Conditional_UNet(5, precision="bf16").to(dev).train()
g = torch.Generator().manual_seed(1000)
x = (torch.rand((32, 3, 256, 256), generator=g) * 2 - 1).to(dev)
c = torch.eye(5)[torch.arange(32) % 5].to(dev)
reducer = None


def build(tail_mb, bucket_mb=12.0):
    global reducer
    if reducer is not None:
        reducer.remove_hooks()
    reducer = GradBucketReducer(ready_order(net), bucket_mb=bucket_mb, ready_order=True, tail_mb=tail_mb).attach(net)
    reducer.world = 2                  # take the collective path
    reducer._avg_op = True             # as on RCCL (native AVG: no division pass)


build(None)
ddp.dist.all_reduce = fake_all_reduce
opt = torch.optim.Adam(list(net.parameters()), lr=1e-4, betas=(0.0, 0.999), weight_decay=1e-4 / 20, fused=True)
UG.prepare_side_stream(dev)


def step():
    reducer.zero_grad()
    out = net(x, c)
    loss = ops.l1_loss(out, x)
    loss.backward()
    reducer.finalize()
    opt.step()


def timed(steps=20, warm=5):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return marks[0].elapsed_time(marks[steps]) / steps, per[len(per) // 2]


VARIANTS = [
    ("no collective kernel (events only)", dict(grid=0, usec=0), 0, None),
    ("32 WG x 200 us", dict(grid=32, usec=200), 0, None),
    ("64 WG x 200 us", dict(grid=64, usec=200), 0, None),
    ("32 WG x 400 us", dict(grid=32, usec=400), 0, None),
    ("32 WG x 200 us, persistent grids on 224 CUs", dict(grid=32, usec=200), 224, None),
    ("no collective kernel, persistent grids on 224 CUs", dict(grid=0, usec=0), 224, None),
    ("32 WG x (25 us + bytes / 70 GB/s)", dict(grid=32, usec=-1), 0, None),
    ("32 WG x (25 us + bytes / 70 GB/s), tail bucket 1 MiB", dict(grid=32, usec=-1), 0, 1.0),
    ("no collective kernel, tail bucket 1 MiB", dict(grid=0, usec=0), 0, 1.0),
]
if len(sys.argv) > 1 and sys.argv[1] == "buckets":       # bucket-size sweep under the duration model, tail bucket on
    VARIANTS = [("no collective kernel (events only)", dict(grid=0, usec=0), 0, (1.0, 12.0))]
    for bmb in (4.0, 8.0, 12.0, 16.0, 32.0):
        for tail in (0.25, 1.0, 4.0):
            VARIANTS.append((f"model, bucket_mb {bmb:g}, tail_mb {tail:g}", dict(grid=32, usec=-1), 0, (tail, bmb)))
if len(sys.argv) > 1 and sys.argv[1] == "sensitivity":   # the two layouts under a slower / faster collective than the 70 GB/s of the model
    VARIANTS = [("no collective kernel (events only)", dict(grid=0, usec=0), 0, (1.0, 12.0))]
    for gbs in (35.0, 70.0, 140.0):
        for bmb, tail in ((12.0, 1.0), (32.0, 1.0), (32.0, 2.0), (32.0, 4.0)):
            VARIANTS.append((f"25 us + bytes / {gbs:g} GB/s, bucket_mb {bmb:g}, tail_mb {tail:g}", dict(grid=32, usec=-1, gbs=gbs), 0, (tail, bmb)))
for rnd in range(2):
    for name, cfg, cus, tail in VARIANTS:
        build(*tail) if isinstance(tail, tuple) else build(tail)
        CFG.update(cfg)
        _lib.call("wu_set_option", 10, cus)
        LOG.clear()
        step()
        torch.cuda.synchronize()
        layout = "+".join(str(b["flat"].numel() * 4 // 1024) for b in reducer.buckets)
        mean, med = timed()
        print(f"round {rnd}: {name:56s} {mean:7.3f} ms/step (median {med:7.3f})   buckets KiB {layout}, launch order {reducer.launch_log}", flush=True)
_lib.call("wu_set_option", 10, 0)
