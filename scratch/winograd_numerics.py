"""VERDICT r3 next #7, the numerics half of the go / no-go (CPU, no kernel needed): Winograd F(2x2, 3x3) for the `up1.0`-shaped conv
(192 -> 64, cunet.py:78) with bf16 matrix operands -- transformed inputs B^T d B and transformed weights G g G^T rounded to bf16 (what an
MFMA would consume), fp32 transforms and fp32 accumulation -- against the direct conv with bf16 operands (what the production kernel
computes) and against fp32.  Reports forward max-abs / relative error and the cosine of the data- and weight-gradient computed THROUGH
the Winograd form (its transpose uses the same rounded operands)."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
rb = lambda t: t.to(torch.bfloat16).float()


class RB(torch.autograd.Function):           # bf16 rounding with a straight-through gradient (operand rounding only)
    @staticmethod
    def forward(ctx, t):
        return rb(t)

    @staticmethod
    def backward(ctx, g):
        return g


def winograd(x, w, round_ops):
    n, c, h, wd = x.shape
    k = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                     # n, c, th, tw, 4, 4
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)        # B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)                  # G g G^T
    if round_ops:
        V, U = RB.apply(V), RB.apply(U)
    M = torch.einsum("kcim,nctuim->nktuim", U, V)
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, M, AT)            # n, k, th, tw, 2, 2
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, h, wd)


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


n, c, k, s = 2, 192, 64, 64
# the decoder's concat input: ReLU outputs (skip half) and AdaIN / dropout outputs (upsampled half), O(1); kaiming-uniform weights
x = torch.cat([torch.randn(n, 128, s, s) * (torch.rand(n, 128, s, s) > 0.3) / 0.7, torch.relu(torch.randn(n, 64, s, s))], 1)
w = (torch.rand(k, c, 3, 3) * 2 - 1) / (c * 9) ** 0.5 * 3 ** 0.5
b = torch.zeros(k)
x, w = rb(x), w                                                    # activations are stored in bf16; weights are fp32 parameters
gy = rb(torch.randn(n, k, s, s))
res = {}
for name in ("fp32 direct", "bf16 direct", "bf16 winograd", "fp32 winograd"):
    xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if name == "fp32 direct":
        y = F.conv2d(xi, wi, b, padding=1)
    elif name == "bf16 direct":
        y = F.conv2d(xi, RB.apply(wi), b, padding=1)
    else:
        y = winograd(xi, wi, name.startswith("bf16"))
    y.backward(gy)
    res[name] = (y.detach(), xi.grad.clone(), wi.grad.clone())
ref = res["fp32 direct"]
scale = ref[0].abs().max().item()
print(f"conv 192 -> 64, {s}x{s}, B={n}: |y|max {scale:.3f}")
for name in ("bf16 direct", "bf16 winograd", "fp32 winograd"):
    y, dx, dw = res[name]
    e = (y - ref[0]).abs()
    print(f"{name:14s}: forward max-abs {e.max().item():.3e} ({e.max().item() / scale:.2e} of scale), rms {e.pow(2).mean().sqrt().item():.3e}; "
          f"after bf16 storage rounding max-abs {(rb(y) - rb(ref[0])).abs().max().item():.3e}; dgrad cos {cos(dx, ref[1]):.7f}, wgrad cos {cos(dw, ref[2]):.7f}")
yw, yd = res["bf16 winograd"][0], res["bf16 direct"][0]
print(f"error ratio winograd / direct (rms): {((yw - ref[0]).pow(2).mean().sqrt() / (yd - ref[0]).pow(2).mean().sqrt()).item():.2f}")
