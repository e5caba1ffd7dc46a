"""adain_upcat_bwd (B=32, three levels): the tree's library against scratch/_oldlib/libwu_old.so, same box, interleaved, outputs compared."""
import ctypes, os, statistics, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch
from wu import _lib, kernels as K
from wu.layout import empty_nhwc
new = _lib.load()
old = ctypes.CDLL(os.path.join(ROOT, "scratch", "_oldlib", "libwu_old.so"))
for name, (res, args) in _lib.SIGNATURES.items():
    if hasattr(old, name):
        f = getattr(old, name); f.restype, f.argtypes = res, args
dev, bf, B = torch.device("cuda:0"), torch.bfloat16, 32
def act(c, s): return (torch.rand((B, s, s, c), device=dev) * 2 - 1).to(bf).permute(0, 3, 1, 2)
def timed(fn, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for (c, h, cs) in [(128, 128, 64), (256, 64, 128), (512, 32, 256)]:
    x = act(c, h); cat = empty_nhwc(B, c + cs, 2 * h, 2 * h, bf, dev); gc = act(c + cs, 2 * h)
    ys = torch.rand((B, c), device=dev) + 0.5; ym = torch.rand((B, c), device=dev)
    st = K.adain_stats(x, 1e-5); mb = K.adain_upcat(x, st, ys, ym, cat, 0.3, 123, True)
    outs = {}
    def run(lib, tag):
        orig = _lib._lib; _lib._lib = lib
        try:
            dx = empty_nhwc(B, c, h, h, bf, dev)
            d = K.adain_upcat_bwd(gc, x, st, ys, dx, 0.3, 123, mb, 1)
            outs[tag] = (dx, d)
        finally:
            _lib._lib = orig
    res = {"old": [], "new": []}
    for _ in range(7):
        for tag, lib in (("old", old), ("new", new)):
            res[tag].append(timed(lambda: run(lib, tag)))
    rel = ((outs["new"][0].float() - outs["old"][0].float()).norm() / outs["old"][0].float().norm()).item()
    print(f"C={c} {2*h}->{h}: old {statistics.median(res['old']):7.1f} us   new {statistics.median(res['new']):7.1f} us   dx rel diff {rel:.1e}")
