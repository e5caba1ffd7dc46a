"""Does SNDisc's forward give the same numbers when it runs on a second stream beside the estimator's forward?  D in eval mode (a pure
function of its input): outputs [out, c1..c4] on the main stream alone (reference), on the side stream alone, and on the side stream
while the ResNet-101 estimator runs on the main stream -- the arrangement of WeatherTransferStep.update_inference."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

import disc  # noqa: E402
from wu.resnet import ResNet101Estimator  # noqa: E402
from wu.unet_graph import _side_stream  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
train = len(sys.argv) > 2 and sys.argv[2] == "train"
torch.manual_seed(3)
D = disc.SNDisc(5, precision="bf16").to(dev)
E = ResNet101Estimator(5, precision="bf16").to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(4)
x = (torch.rand((B, 3, 256, 256), generator=g) * 2 - 1).to(dev)
c = torch.softmax(torch.randn(B, 5, generator=g), dim=1).to(dev)
state = {k: v.clone() for k, v in D.state_dict().items()}
D.train(train)
main = torch.cuda.current_stream(dev)
side = _side_stream(dev)


GRAD = len(sys.argv) > 3 and sys.argv[3] == "grad"
xg = x.clone().requires_grad_(True)


def run(where, with_est):
    D.load_state_dict(state)
    torch.cuda.synchronize()
    with torch.enable_grad() if GRAD else torch.no_grad():
        x = xg if GRAD else globals()["x"]
        if where == "main":
            o = D(x, c)
            if with_est:
                E(x)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                o = D(x, c)
            if with_est:
                E(x)
            main.wait_stream(side)
    torch.cuda.synchronize()
    return [t.detach().float().clone() for t in o]


ref = run("main", False)
for name, where, est in (("main again", "main", False), ("main, estimator after", "main", True), ("side alone", "side", False),
                         ("side + estimator on main", "side", True), ("side + estimator on main (2)", "side", True)):
    o = run(where, est)
    diffs = [f"{n}: {(a - b).abs().max().item():.3e}" for n, a, b in zip(("out", "c1", "c2", "c3", "c4"), ref, o) if not torch.equal(a, b)]
    print(f"{name:32s} " + ("bit-identical" if not diffs else "DIFFERS  " + ", ".join(diffs)))
