"""Same-box, interleaved A/B of the conv kernels of TWO library builds: the tree's libwu_kernels.so against scratch/_oldlib/libwu_old.so
(another revision, scratch/build_baseline_lib.sh).  For the 13 MFMA layers of the B=32 256x256 step: forward (as the fused graph
launches it: +pool / +gate bits / plain), data gradient (gate bits where the graph uses them), outputs compared bit for bit.

    python scratch/ab_lib.py [rounds]
"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
import torch  # noqa: E402

from wu import _lib, kernels as K  # noqa: E402
from wu.layout import empty_nhwc  # noqa: E402


def load_old():
    lib = ctypes.CDLL(os.path.join(ROOT, "scratch", "_oldlib", "libwu_old.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    new, old = _lib.load(), load_old()
    for kv in sys.argv[2:]:                  # options for the NEW library only, e.g. 0=2 (always the 4-wave conv shape)
        k, v = kv.split("=")
        new.wu_set_option(int(k), int(v))
    dev, bf, B, S = torch.device("cuda:0"), torch.bfloat16, 32, 256
    sp = torch.cuda.current_stream().cuda_stream

    def act(c, s, lo=-1.0):
        return (torch.rand((B, s, s, c), device=dev) * (1 - lo) + lo).to(bf).permute(0, 3, 1, 2)

    def timed(fn, reps=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    layers = [("down1.2", 64, 64, S, "pool"), ("down2.0", 64, 128, S // 2, "bits"), ("down2.2", 128, 128, S // 2, "pool"),
              ("down3.0", 128, 256, S // 4, "bits"), ("down3.2", 256, 256, S // 4, "pool"), ("down4.0", 256, 512, S // 8, "bits"),
              ("down4.2", 512, 512, S // 8, "plain"), ("up3.0", 768, 256, S // 4, "bits"), ("up3.2", 256, 256, S // 4, "plain"),
              ("up2.0", 384, 128, S // 2, "bits"), ("up2.2", 128, 128, S // 2, "plain"), ("up1.0", 192, 64, S, "bits"), ("up1.2", 64, 64, S, "plain")]
    tot = {"fo": 0.0, "fn": 0.0, "do": 0.0, "dn": 0.0, "wo": 0.0, "wn": 0.0}
    print(f"{'layer':8s} {'shape':16s} {'fwd old':>8s} {'fwd new':>8s} {'ratio':>6s}   {'dgrad old':>9s} {'dgrad new':>9s} {'ratio':>6s}   {'wgrad old':>9s} {'wgrad new':>9s} {'ratio':>6s}  bitwise")
    for name, ci, co, s, kind in layers:
        x, gy = act(ci, s), act(co, s)
        w = (torch.rand((co, ci, 3, 3), device=dev) * 2 - 1) * 0.05
        wf, wd = K.pack_conv3x3(w, 1)
        bias = torch.rand(co, device=dev) - 0.5
        ys = [empty_nhwc(B, co, s, s, bf, dev) for _ in range(2)]
        dxs = [empty_nhwc(B, ci, s, s, bf, dev) for _ in range(2)]
        pls = [empty_nhwc(B, co, s // 2, s // 2, bf, dev) for _ in range(2)]
        nbits = new.wu_gate_bits_bytes(B, s, s, co) // 4
        bits = [torch.empty(nbits, dtype=torch.int32, device=dev) for _ in range(2)]
        # the gate of the data gradient belongs to the conv's INPUT tensor (ci channels): bits of a ReLU output of that shape
        gb_in = torch.randint(-2**31, 2**31 - 1, (new.wu_gate_bits_bytes(B, s, s, ci) // 4,), dtype=torch.int32, device=dev)
        ldx, ldy, ldg, lddx = K.nhwc_ld(x), K.nhwc_ld(ys[0]), K.nhwc_ld(gy), K.nhwc_ld(dxs[0])

        def fwd(lib, i):
            if kind == "pool":
                rc = lib.wu_conv3x3_relu_pool_fwd(x.data_ptr(), ldx, wf.data_ptr(), bias.data_ptr(), ys[i].data_ptr(), ldy, pls[i].data_ptr(),
                                                  K.nhwc_ld(pls[i]), B, s, s, ci, co, 1, sp)
            elif kind == "bits":
                rc = lib.wu_conv3x3_fwd_bits(x.data_ptr(), ldx, wf.data_ptr(), bias.data_ptr(), ys[i].data_ptr(), ldy, bits[i].data_ptr(), None,
                                             B, s, s, ci, co, 1, 1, sp)
            else:
                rc = lib.wu_conv3x3_fwd(x.data_ptr(), ldx, wf.data_ptr(), bias.data_ptr(), ys[i].data_ptr(), ldy, B, s, s, ci, co, 1, 1,
                                        None, 0, 0, None, 0, 0, 1, sp)
            assert rc == 0, lib.wu_last_error()

        def dgrad(lib, i):
            # gy (co ch) -> dx (ci ch), gated by the bits of the block's first-conv output where the graph has them (second convs)
            if kind in ("pool", "plain"):
                rc = lib.wu_conv3x3_fwd_bits(gy.data_ptr(), ldg, wd.data_ptr(), None, dxs[i].data_ptr(), lddx, None, gb_in.data_ptr(),
                                             B, s, s, co, ci, 0, 1, sp)
            else:
                rc = lib.wu_conv3x3_fwd(gy.data_ptr(), ldg, wd.data_ptr(), None, dxs[i].data_ptr(), lddx, B, s, s, co, ci, 1, 0,
                                        None, 0, 0, None, 0, 0, 1, sp)
            assert rc == 0, lib.wu_last_error()

        dws = [torch.empty((co, ci, 3, 3), device=dev) for _ in range(2)]
        dbs = [torch.empty((co,), device=dev) for _ in range(2)]
        ws = K.workspace(new.wu_conv3x3_wgrad_workspace(B, s, s, ci, co, 1, 1), dev)

        def wgrad(lib, i):
            rc = lib.wu_conv3x3_wgrad(x.data_ptr(), ldx, gy.data_ptr(), ldg, None, 0, 0, dws[i].data_ptr(), dbs[i].data_ptr(), ws.data_ptr(), ws.numel(),
                                      B, s, s, ci, co, 1, 0, 1, sp)
            assert rc == 0, lib.wu_last_error()

        for lib, i in ((old, 0), (new, 1)):
            fwd(lib, i); dgrad(lib, i); wgrad(lib, i)
        torch.cuda.synchronize()
        same = torch.equal(ys[0], ys[1]) and torch.equal(dxs[0], dxs[1]) and (kind != "pool" or torch.equal(pls[0], pls[1])) and \
            (kind != "bits" or torch.equal(bits[0], bits[1])) and torch.equal(dws[0], dws[1]) and torch.equal(dbs[0], dbs[1])
        r = {k: [] for k in tot}
        for _ in range(rounds):
            r["fo"].append(timed(lambda: fwd(old, 0)))
            r["fn"].append(timed(lambda: fwd(new, 1)))
            r["do"].append(timed(lambda: dgrad(old, 0)))
            r["dn"].append(timed(lambda: dgrad(new, 1)))
            r["wo"].append(timed(lambda: wgrad(old, 0)))
            r["wn"].append(timed(lambda: wgrad(new, 1)))
        m = {k: statistics.median(v) for k, v in r.items()}
        for k in tot:
            tot[k] += m[k]
        print(f"{name:8s} {f'{ci}->{co} @{s}':16s} {m['fo']:8.1f} {m['fn']:8.1f} {m['fn'] / m['fo']:6.3f}   {m['do']:9.1f} {m['dn']:9.1f} {m['dn'] / m['do']:6.3f}   "
              f"{m['wo']:9.1f} {m['wn']:9.1f} {m['wn'] / m['wo']:6.3f}  {same}")
        del x, gy, ys, dxs, pls, bits, dws, dbs
    print(f"{'sum':8s} {'':16s} {tot['fo']:8.1f} {tot['fn']:8.1f} {tot['fn'] / tot['fo']:6.3f}   {tot['do']:9.1f} {tot['dn']:9.1f} {tot['dn'] / tot['do']:6.3f}   "
          f"{tot['wo']:9.1f} {tot['wn']:9.1f} {tot['wn'] / tot['wo']:6.3f}")


if __name__ == "__main__":
    main()
