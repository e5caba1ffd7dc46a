"""ctypes binding of libwu_kernels.so -- the C ABI declared in include/wu_kernels.h.

There is deliberately NO fallback: if the shared library is missing or a call fails, the product
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes
import os

from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_uint64, c_void_p, POINTER

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libwu_kernels.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
OPT_CONV_V2, OPT_CONV_PERSISTENT, OPT_WGRAD_V2 = 0, 1, 2
FAM_CONV_FWD, FAM_WGRAD, FAM_CONV_DGRAD, FAM_CONV_S2, FAM_WGRAD_S2, FAM_CONV1X1 = 1, 2, 3, 4, 5, 6
FAMILY_KERNEL = {1: "conv3x3_mfma_v2_kernel (forward + data-gradient convs; generic conv3x3_mfma_kernel for fp32 / narrow images)",
                 2: "conv3x3_wgrad_v2_kernel (generic conv3x3_wgrad_kernel for fp32 / narrow images)",
                 3: "conv3x3_mfma_kernel<T,1,true> (in-kernel gated dgrad; unused by the fused graph)",
                 4: "stride-2 3x3 conv forward + data gradient: conv1x1_pw3_kernel<.., CONV> (gathered rows); conv3x3_mfma_kernel<T,2,false> / tap-list classes for 64-cout and fp32",
                 5: "conv3x3_wgrad_kernel<T,2>",
                 6: "pointwise convs (estimator): conv1x1_pw3_kernel; conv1x1_mfma_kernel<T> for strided / small / fp32"}

P, I, F, U64, SZ = c_void_p, c_int, c_float, c_uint64, c_size_t

# name -> (restype, argtypes); mirrors include/wu_kernels.h one to one
SIGNATURES = {
    "wu_last_error": (c_char_p, []),
    "wu_version": (I, []),
    "wu_cu_count": (I, []),
    "wu_set_option": (I, [I, I]),
    "wu_set_debug_buffer": (I, [P]),
    "wu_stream_create_cu_mask": (I, [P, I, P]),
    "wu_stream_destroy": (I, [P]),
    "wu_spectral_norm_scratch_floats": (SZ, [I, I]),
    "wu_spectral_norm_fwd": (I, [P, I, I, P, P, I, F, P, P, P, P]),
    "wu_spectral_norm_bwd": (I, [P, P, P, P, P, P, I, I, P, P]),
    "wu_event_create": (I, [P]),
    "wu_event_destroy": (I, [P]),
    "wu_stream_order_after": (I, [P, P, P]),
    "wu_spectral_norm_fwd_multi": (I, [I, P, P, P, P, P, I, F, P, P, P, P, P, P]),
    "wu_spectral_norm_bwd_multi": (I, [I, P, P, P, P, P, P, P, P, P, P]),
    "wu_pack_conv3x3": (I, [P, P, P, I, I, P, I, P]),
    "wu_pack_conv3x3_multi": (I, [I, P, P, P, P, P, I, P]),
    "wu_conv3x3_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, P, I, I, P, I, I, I, P]),
    "wu_conv3x3_gate_bits_supported": (I, [I, I, I, I, I, I, I]),
    "wu_gate_bits_bytes": (SZ, [I, I, I, I]),
    "wu_conv3x3_fwd_bits": (I, [P, I, P, P, P, I, P, P, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_relu_pool_fwd": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_relu_pool_bits_fwd": (I, [P, I, P, P, P, I, P, I, P, P, I, I, I, I, I, I, P]),
    "wu_conv3x3_small_supported": (I, [I, I, I, I, I, I, I, I]),
    "wu_conv3x3_small_fwd": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_relu_head_supported": (I, [I, I, I, I, I, I, I]),
    "wu_conv3x3_relu_head_fwd": (I, [P, I, P, P, P, I, P, P, P, I, I, I, I, I, I, P]),
    "wu_maxpool2_bwd_bits": (I, [P, P, P, I, P, I, P, I, I, I, I, I, I, P]),
    "wu_conv3x3_wgrad_workspace": (SZ, [I, I, I, I, I, I, I]),
    "wu_conv3x3_wgrad": (I, [P, I, P, I, P, I, I, P, P, P, SZ, I, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_s2_dgrad_workspace": (SZ, [I, I, I, I, I]),
    "wu_conv3x3_s2_dgrad": (I, [P, I, P, I, I, P, P, I, P, SZ, P, I, I, I, I, I, I, I, I, P]),
    "wu_act_gate": (I, [P, I, P, I, P, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_c3_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "wu_conv3x3_c3_gate_bits_supported": (I, [I, I, I, I, I, P, I]),
    "wu_conv3x3_c3_fwd_bits": (I, [P, P, P, P, P, I, P, I, I, I, I, I, I, P]),
    "wu_conv3x3_c3_wgrad": (I, [P, P, I, I, P, I, I, P, P, P, SZ, I, I, I, I, I, I, I, P]),
    "wu_thin_workspace_bytes": (SZ, []),
    "wu_conv3x3_c3_dgrad": (I, [P, I, I, P, I, I, P, P, P, I, I, I, I, I, I, I, P]),
    "wu_conv1x1_tanh_fwd": (I, [P, I, P, P, P, I, I, I, I, I, P]),
    "wu_conv1x1_tanh_bwd": (I, [P, P, P, I, P, P, I, P, P, P, SZ, I, I, I, I, I, I, I, P]),
    "wu_maxpool2_fwd": (I, [P, I, P, I, I, I, I, I, I, P]),
    "wu_maxpool2_bwd": (I, [P, I, P, I, P, I, P, I, I, I, I, I, I, I, P]),
    "wu_adain_style_fwd": (I, [P, P, P, F, P, P, P, I, I, I, P]),
    "wu_adain_style_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "wu_adain_style_fwd_multi": (I, [I, P, P, P, P, P, P, P, I, P, I, P]),
    "wu_adain_style_bwd_multi": (I, [I, P, P, P, P, P, P, P, P, I, P, I, I, P]),
    "wu_adain_stats": (I, [P, I, P, P, I, I, I, I, F, I, P]),
    "wu_adain_upcat_fwd": (I, [P, I, P, P, P, P, I, I, I, I, I, F, U64, P, P, I, I, P]),
    "wu_adain_upcat_bwd": (I, [P, I, P, I, P, P, P, I, P, P, P, P, I, I, I, I, F, U64, P, I, I, P]),
    "wu_dropout_mask": (I, [P, I, I, I, I, F, U64, P]),
    "wu_l1_mean_scratch_floats": (SZ, []),
    "wu_l1_mean": (I, [P, P, P, P, P, ctypes.c_longlong, P]),
    "wu_sumpool_fwd": (I, [P, I, P, I, I, I, I, I, P]),
    "wu_sumpool_bwd": (I, [P, P, I, I, I, I, I, I, P]),
    "wu_nhwc_to_nchw_f32": (I, [P, I, P, I, I, I, I, I, P]),
    "wu_nchw_f32_to_nhwc": (I, [P, P, I, I, I, I, I, I, P]),
    "wu_conv1x1_fwd": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, I, I, I, P]),
    "wu_conv1x1_chain_supported": (I, [I, I, I, I]),
    "wu_conv1x1_chain": (I, [P, I, P, P, P, I, I, P, I, I, P, I, P, P, I, P, I, I, P, I, ctypes.c_longlong, I, I, I, I, P]),
    "wu_stem7x7_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "wu_stem7x7_dgrad": (I, [P, I, P, P, I, I, I, I, I, P]),
    "wu_maxpool3s2_fwd": (I, [P, I, P, I, P, I, I, I, I, I, P]),
    "wu_maxpool3s2_bwd": (I, [P, I, P, P, I, P, I, I, I, I, I, I, I, P]),
    "wu_image_geo_bytes": (SZ, []),
    "wu_image_workspace_bytes": (SZ, [I, I, I]),
    "wu_image_geometry": (I, [P, P, P, SZ, P, P, I, I, I, I, P]),
    "wu_image_color_jitter": (I, [P, P, P, P, I, I, P]),
    "wu_prof_begin": (I, [ctypes.c_uint, I]),
    "wu_prof_query": (I, [I, POINTER(c_int), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "wu_prof_end": (I, []),
}

_lib = None
LOADED_PATH = None       # the file load() actually opened (bench.py reports it when it is not the in-tree library)


def load():
    """Load the shared library (once).  Raises if it has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python weather-unet_amd/wu/_build.py` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback for the hot path.")
        # torch bundles its own libamdhip64 (same soname as /opt/rocm's): import it FIRST so the dynamic
        # loader binds this library to the HIP runtime instance torch initialises -- two runtime copies in
        # one process do not share devices, streams or allocations.
        import torch  # noqa: F401
        # WU_AB_LIB: benchmark harness only (scratch/): run the whole product path on ANOTHER build of the library (a baseline
        # revision, scratch/build_baseline_lib.sh) for same-box A/B timing of the full step; entry points that build lacks stay unbound
        ab = os.environ.get("WU_AB_LIB")
        if not ab:
            _check_not_stale()
        else:
            # never silently: a leftover variable would run training on a stale baseline build (advisor, round 3)
            import sys
            print(f"[wu] WARNING: WU_AB_LIB is set -- running on {ab} instead of {LIB_PATH}; no staleness check, entry points "
                  "that build lacks stay unbound.  Benchmark A/B harness only (scratch/ab_lib.py).", file=sys.stderr, flush=True)
        global LOADED_PATH
        LOADED_PATH = ab or LIB_PATH
        lib = ctypes.CDLL(LOADED_PATH)
        for name, (res, args) in SIGNATURES.items():
            if ab and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        # WU_SET_OPTIONS="key=value,...": diagnostic switches of the library (include/wu_kernels.h, wu_set_option) from the environment --
        # same-box A/B runs of a whole benchmark (scratch/ab_gan_env.sh); announced, never silent
        opts = os.environ.get("WU_SET_OPTIONS")
        if opts:
            import sys
            print(f"[wu] WU_SET_OPTIONS={opts}", file=sys.stderr, flush=True)
            for kv in opts.split(","):
                k, _, v = kv.partition("=")
                if lib.wu_set_option(int(k), int(v)) != 0:
                    raise RuntimeError(f"WU_SET_OPTIONS: wu_set_option({k}, {v}) refused")
    return _lib


def _check_not_stale():
    """A library built from other sources than the ones in the tree must never run silently: compare the content hash the
    build recorded with the tree's (skipped when the sources did not travel with the library)."""
    from . import _build
    if not os.path.isdir(_build.CSRC):
        return
    if _build.is_stale():
        if os.environ.get("WU_ALLOW_STALE_LIB"):
            return
        raise RuntimeError(f"{LIB_PATH} was not built from the sources in {_build.CSRC} (content hash mismatch): "
                           "run `python weather-unet_amd/wu/_build.py` (or __graft_entry__.build()); "
                           "WU_ALLOW_STALE_LIB=1 overrides")


def check(rc, what):
    if rc != 0:
        msg = load().wu_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero return."""
    check(getattr(load(), name)(*args), name)


def prof_begin(families, max_launches):
    mask = 0
    for f in families:
        mask |= 1 << f
    call("wu_prof_begin", mask, int(max_launches))


def prof_query(family):
    n, ms, fl, by = c_int(0), c_double(0), c_double(0), c_double(0)
    call("wu_prof_query", family, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
    return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}


def prof_end():
    call("wu_prof_end")
