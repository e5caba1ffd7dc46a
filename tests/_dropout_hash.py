"""numpy restatement of the dropout counter hash (splitmix64 finaliser of seed * golden + group + constant) (weather-unet_amd/csrc/wu_common.h: wu_mix64 / wu_rand4, csrc/glue.hip: keep_thr) -- test
infrastructure: the CPU test pins its statistics, the GPU test pins the kernels' masks to it bit for bit."""
import numpy as np

_LO, _S32 = np.uint64(0xFFFFFFFF), np.uint64(32)


def mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def rand4(seed, group):
    """64 bits per GROUP of four consecutive NHWC elements: element e of the group keeps iff bits [16 e, 16 e + 16) < keep_thr(p)."""
    with np.errstate(over="ignore"):
        g = np.asarray(group, dtype=np.uint64)
        return mix64(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + g + np.uint64(0x632BE59BD9B4E019))


def rand4_cheap(seed, group):
    """The cheaper candidate of round 4 (three 32 x 32 -> 64 multiply-and-fold rounds behind a once-per-kernel splitmix64 of the seed): built,
    statistically as good as the shipped hash (same test), and NOT faster -- the fused AdaIN / upsample / dropout kernel is not bound by its
    draws (158.7 / 102.0 / 45.8 us against 162.4 / 98.4 / 45.1 for the three levels) -- so the shipped hash stayed.  Kept as the record."""
    with np.errstate(over="ignore"):
        s = mix64(np.array([seed], dtype=np.uint64) + np.uint64(0x632BE59BD9B4E019))[0]
        g = np.asarray(group, dtype=np.uint64)
        a, b = (g ^ s) & _LO, ((g >> _S32) ^ (s >> _S32)) & _LO
        k0, k1 = np.uint64(0x53c5ca59), np.uint64(0x74743c1b)
        c = (a ^ k0) * (b ^ k1); a, b = c & _LO, c >> _S32
        c = (a ^ k0) * (b ^ k1); a, b = c & _LO, c >> _S32
        lo = a ^ b
        c = (a ^ np.uint64(0x9E3779B9)) * (b ^ np.uint64(0x85EBCA6B))
        hi = (c & _LO) ^ (c >> _S32)
        return lo | (hi << _S32)


def keep_thr(p):
    return 0x10000 if p <= 0 else int((1.0 - float(np.float32(p))) * 65536.0 + 0.5)


def keep_mask_nchw(n, c, h2, w2, p, seed):
    """What wu_dropout_mask writes: (N, C, H2, W2) uint8, the RNG's index space being the NHWC linear index."""
    i = np.arange(n * h2 * w2 * c, dtype=np.uint64)
    r = rand4(seed, i >> np.uint64(2))
    u = (r >> (np.uint64(16) * (i & np.uint64(3)))) & np.uint64(0xFFFF)
    keep = (u < np.uint64(keep_thr(p))).astype(np.uint8)
    return keep.reshape(n, h2, w2, c).transpose(0, 3, 1, 2)
