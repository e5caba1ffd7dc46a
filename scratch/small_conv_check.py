"""Small-image 3x3 conv kernel (csrc/conv3x3_small.hip, option 3) against the generic template on the estimator's shapes: time inside a captured
graph (20 launches per replay, median of 7), max-abs difference between the two kernels' bf16 outputs, and both against an fp32 CPU conv."""
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "weather-unet_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from wu import kernels as K, _lib  # noqa: E402
from wu.layout import empty_nhwc, as_nhwc  # noqa: E402

dev = torch.device("cuda:0")


def graph_time(fn, n=20, reps=7):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


torch.manual_seed(5)
cases = [("est 256 @16 B32", 32, 256, 256, 16, 16, False), ("est 256 @16 B64", 64, 256, 256, 16, 16, False), ("est 256 @16 B128", 128, 256, 256, 16, 16, False),
         ("est 512 @8 B32", 32, 512, 512, 8, 8, False), ("est 512 @8 B64", 64, 512, 512, 8, 8, False), ("est 512 @8 B128", 128, 512, 512, 8, 8, False),
         ("est dgrad 256 @16 B32 gated", 32, 256, 256, 16, 16, True), ("est dgrad 512 @8 B33 gated", 33, 512, 512, 8, 8, True),
         ("ragged 64->128 @13x9 B3", 3, 64, 128, 13, 9, True), ("ragged 96->64 @5x3 B7", 7, 96, 64, 5, 3, False), ("64->64 @4x4 B9", 9, 64, 64, 4, 4, False),
         ("128->128 @24x16 B2", 2, 128, 128, 24, 16, True)]
check_cpu = "--no-cpu" not in sys.argv
for name, B, ci, co, h, w, gated in cases:
    x = as_nhwc((torch.rand((B, ci, h, w), device=dev) - 0.3), _lib.BF16)
    wt = (torch.rand((co, ci, 3, 3), device=dev) - 0.5) * (6.0 / (9 * ci)) ** 0.5
    wf, _ = K.pack_conv3x3(wt, _lib.BF16)
    b = torch.rand(co, device=dev) - 0.5
    eg = as_nhwc(torch.rand((B, co, h, w), device=dev) - 0.5, _lib.BF16) if gated else None
    act = K.ACT_NONE if gated else K.ACT_RELU
    res = {}
    for opt in (1, 0):
        _lib.call("wu_set_option", 3, 2 if opt else 0)
        y = torch.full((B, h, w, co), float("nan"), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
        fn = lambda: K.conv3x3(x, wf, None if gated else b, y, 1, act, egate=eg, egate_act=K.ACT_RELU if gated else K.ACT_NONE)
        t = graph_time(fn)
        res[opt] = (t, y.float().clone())
    _lib.call("wu_set_option", 3, 1)
    wc = K.chunk_major(wf)
    yc = torch.full((B, h, w, co), float("nan"), dtype=torch.bfloat16, device=dev).permute(0, 3, 1, 2)
    tc = graph_time(lambda: K.conv3x3_small(x, wc, None if gated else b, yc, act, egate=eg, egate_act=K.ACT_RELU if gated else K.ACT_NONE))
    same = torch.equal(yc.float(), res[1][1])
    fl = 2.0 * B * h * w * 9 * ci * co
    d = (res[1][1] - res[0][1]).abs().max().item()
    line = f"{name:30s} chunk-major {tc:7.1f} us {fl / tc / 1e6:6.0f} TF/s ({'bit-identical' if same else 'DIFFERS'}) | small {res[1][0]:7.1f} us {fl / res[1][0] / 1e6:6.0f} TF/s | generic {res[0][0]:7.1f} us {fl / res[0][0] / 1e6:6.0f} TF/s | max|small-generic| {d:.3e}"
    if check_cpu and fl < 3e10:
        ref = F.conv2d(x.float().cpu(), wt.to(torch.bfloat16).float().cpu(), None if gated else b.cpu(), padding=1)
        ref = ref * (eg.float().cpu() > 0) if gated else torch.relu(ref)
        e1 = (res[1][1].cpu() - ref).abs().max().item()
        e0 = (res[0][1].cpu() - ref).abs().max().item()
        line += f" | vs fp32 conv: small {e1:.3e} generic {e0:.3e} (scale {ref.abs().max().item():.2f})"
    assert not torch.isnan(res[1][1]).any(), name
    print(line, flush=True)
