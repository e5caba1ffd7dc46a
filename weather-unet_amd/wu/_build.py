"""Build libwu_kernels.so for gfx950 with hipcc (cross-compiles without a GPU).

    python weather-unet_amd/wu/_build.py [--force]

The .so is kept in-tree (weather-unet_amd/lib/, git-ignored) so it travels with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libwu_kernels.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
# -packed-fp32-ops: no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in the device code.  Round 3 measured packed-FP32 results of one
# kernel corrupted (single 16-lane groups, a few per launch) while it shared SIMDs with the MFMA waves of ANOTHER kernel running on a
# second stream (csrc/thin.hip: fmac1); the kernels are HBM- or MFMA-bound, the packed forms bought nothing measurable
# (profiles/r03_packed_fp32_ab.txt).  WU_PACKED_FP32=1 builds with them again (A/B work only).
NO_PACKED_FP32 = [] if os.environ.get("WU_PACKED_FP32") == "1" else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-inline-asm"] + NO_PACKED_FP32 + ["-I", INCLUDE]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


STAMP = os.path.join(LIBDIR, "source_hash.txt")


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "wu_kernels.h")]


def source_hash(src=None):
    """sha256 over the compile flags, every header and `src` (or all sources): what an object / the library was built FROM.
    Content, not mtimes: a checkout, a copy to another box or a touched file cannot make a stale object look fresh."""
    h = hashlib.sha256(" ".join(FLAGS[:-2]).encode())
    for f in _headers() + ([src] if src else sources()):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stamps():
    out = {}
    if os.path.exists(STAMP):
        for line in open(STAMP):
            k, _, v = line.strip().partition(" ")
            if v:
                out[k] = v
    return out


def is_stale():
    """True if libwu_kernels.so is missing or was not built from the sources in the tree."""
    return not os.path.exists(LIB) or _stamps().get("libwu_kernels.so") != source_hash()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    stamps = _stamps()
    objs, jobs, new = [], [], {}
    for src in sources():
        name = os.path.basename(src)[:-4] + ".o"
        obj = os.path.join(LIBDIR, name)
        objs.append(obj)
        new[name] = source_hash(src)
        if force or not os.path.exists(obj) or stamps.get(name) != new[name]:
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # the HOST pass of a .hip file does not know the AMDGPU feature name and says so once per function set: not a diagnostic
        err = "\n".join(ln for ln in r.stderr.splitlines() if "packed-fp32-ops' is not a recognized feature" not in ln)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    new["libwu_kernels.so"] = source_hash()
    if jobs or not os.path.exists(LIB) or stamps.get("libwu_kernels.so") != new["libwu_kernels.so"]:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(STAMP, "w") as fh:
        for k in sorted(new):
            fh.write(f"{k} {new[k]}\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
