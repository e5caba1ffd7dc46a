"""Which tensors of a full-size GAN iteration are not reproducible?  Runs tests/test_gpu_round3._iteration several times from identical
state and lists every loss / gradient / buffer that differs from the first run.   python scratch/gan_determinism_probe.py est 64 6"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "weather-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_gpu_round3 as T  # noqa: E402

mode, batch, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
from wu import _lib  # noqa: E402
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    _lib.call("wu_set_option", int(k), int(v))
st, images, rand_images = T._full_step(mode, batch)
d_state = {k: v.clone() for k, v in st.discriminator.state_dict().items()}
ref = T._iteration(st, images, rand_images, d_state)
print("overlap", os.environ.get("WU_GAN_OVERLAP", "1"), "losses", ref[0])
prev = ref
for r in range(1, reps):
    cur = T._iteration(st, images, rand_images, d_state)
    same_prev = cur[0] == prev[0] and all(torch.equal(prev[1][k], cur[1][k]) for k in cur[1])
    prev = cur
    print(f"  run {r} vs run {r - 1}: {'identical' if same_prev else 'differs'}; g_loss_adv {cur[0][2]!r}")
    bad = [f"loss[{i}]" for i, (a, b) in enumerate(zip(ref[0], cur[0])) if a != b]
    bad += [k for k in ref[1] if not torch.equal(ref[1][k], cur[1][k])]
    bad += [k for k in ref[2] if not torch.equal(ref[2][k], cur[2][k])]
    print(f"run {r}: {len(bad)} differing" + (": " + ", ".join(bad[:12]) + (" ..." if len(bad) > 12 else "") if bad else ""))
