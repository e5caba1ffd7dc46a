"""Does the LDS-tiled image-layout 3 -> 3 conv give the same numbers when another kernel runs beside it on a second stream?
A blocker on the main stream holds both streams back so that the two kernels start together; compared bitwise with the conv alone."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K, resnet as RN  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402
from wu.unet_graph import _side_stream  # noqa: E402

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.call("wu_set_option", int(k), int(v))
dev = torch.device("cuda:0")
CODE = _lib.BF16
B, S = 64, 256
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((B, 3, S, S), generator=g) * 2 - 1).to(dev)
w33 = ((torch.rand((3, 3, 3, 3), generator=g) - 0.5) * 0.3).to(dev)
b3 = (torch.rand(3, generator=g) - 0.5).to(dev)
ws = ((torch.rand((64, 3, 7, 7), generator=g) - 0.5) * 0.1).to(dev)
bs = (torch.rand(64, generator=g) - 0.5).to(dev)
stem_y = empty_nhwc(B, 64, S // 2, S // 2, torch.bfloat16, dev)
pool_y = empty_nhwc(B, 64, S // 4, S // 4, torch.bfloat16, dev)
amax = torch.empty(B * (S // 4) * (S // 4) * 64, dtype=torch.uint8, device=dev)
gy_stem = (torch.rand((B, S // 2, S // 2, 64), generator=g) - 0.5).to(torch.bfloat16).to(dev).permute(0, 3, 1, 2)
dx_stem = torch.empty_like(x)
gflip = (torch.rand((B, 3, S, S), generator=g) - 0.5).to(dev)
w11 = ((torch.rand((256, 64), generator=g) - 0.5) * 0.1).to(torch.bfloat16).to(dev)
pw_y = empty_nhwc(B, 256, S // 4, S // 4, torch.bfloat16, dev)
x2 = x.clone()
RN.stem7x7(x, ws, bs, stem_y, 1, CODE)
torch.cuda.synchronize()
stem_ref = stem_y.clone()
main, side = torch.cuda.current_stream(dev), _side_stream(dev)

others = {
    "nothing": lambda: None,
    "stem7x7 fwd": lambda: RN.stem7x7(x, ws, bs, stem_y, 1, CODE),
    "maxpool3s2 fwd": lambda: RN.maxpool3s2(stem_y, pool_y, amax),
    "stem7x7 dgrad": lambda: RN.stem7x7_dgrad(gy_stem, ws, dx_stem, CODE),
    "torch elementwise": lambda: torch.add(x, 1.0),
    "conv1x1 64->256": lambda: RN.conv1x1(pool_y, w11, None, pw_y, 1),
    "stem7x7 fwd (other input)": lambda: RN.stem7x7(x2, ws, bs, stem_y, 1, CODE),
}
subjects = {
    "fwd": lambda out: K.conv3x3_c3(x, w33, b3, out, 1, 0, True, CODE),
    "dgrad": lambda out: K.conv3x3_c3_dgrad(gflip, w33, out, 1, CODE, dy_nchw=True),
}
for sname, subj in subjects.items():
    ref = torch.empty_like(x)
    subj(ref)
    torch.cuda.synchronize()
    for oname, other in others.items():
        bad = 0
        worst = 0.0
        for rep in range(10):
            out = torch.full_like(x, float("nan"))
            torch.cuda.synchronize()
            torch.cuda._sleep(3_000_000)               # ~1.5 ms blocker on the main stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                subj(out)
            other()
            main.wait_stream(side)
            torch.cuda.synchronize()
            if oname.startswith("stem7x7 fwd") and not torch.equal(stem_y, stem_ref):
                print("   (the stem's own output differs too)")
            if not torch.equal(out, ref) and bad == 0 and os.environ.get("PATTERN"):
                wrong = (out != ref) | out.isnan()
                print(f"   wrong elements: {wrong.float().mean().item():.4f} of all; NaN left: {out.isnan().float().mean().item():.4f}")
                print("   per image:", wrong.flatten(1).float().mean(1)[:16].tolist())
                print("   per plane:", wrong.float().mean((0, 2, 3)).tolist())
                print("   per row%16:", [round(v, 3) for v in wrong.float().mean((0, 1, 3)).view(-1, 16).mean(0).tolist()])
                print("   per col%64:", [round(v, 3) for v in wrong.float().mean((0, 1, 2)).view(-1, 64).mean(0).tolist()])
                idx = wrong.nonzero()
                print("   first wrong coordinates (n, plane, h, w):", idx[:40].tolist())
                for n_, c_, h_, w_ in idx[:6].tolist():
                    print(f"     got {out[n_, c_, h_, w_].item():+.6f} want {ref[n_, c_, h_, w_].item():+.6f}")
                tiles = wrong.view(B, 3, 16, 16, 4, 64).float().mean((1, 3, 5))      # (n, ty, tx)
                print("   tiles fully wrong / partly / clean:", (tiles == 1).sum().item(), ((tiles > 0) & (tiles < 1)).sum().item(), (tiles == 0).sum().item())
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, (out - ref).abs().nan_to_num(1e9).max().item())
        print(f"img3 {sname:5s} beside {oname:18s}: {bad}/10 runs differ" + (f" (max |diff| {worst:.3e})" if bad else ""))
