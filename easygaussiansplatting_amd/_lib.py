"""ctypes binding of libegs_hip.so (the C ABI declared in include/egs_hip.h).

There is NO fallback: if the shared library is missing or does not export the
expected ABI, importing the op surface raises -- the product path never routes
through the CPU oracle or plain PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libegs_hip.so")
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 9


class EgsPolicy(C.Structure):
    """Mirror of `struct EgsPolicy` (include/egs_hip.h)."""
    _fields_ = [("near_cull", C.c_int32), ("fov_mode", C.c_int32), ("det_eps", C.c_float),
                ("nan_cull", C.c_int32), ("radius_mode", C.c_int32), ("footprint", C.c_int32),
                ("far_cull", C.c_int32), ("maha_floor", C.c_int32), ("alpha_clamp", C.c_int32),
                ("alpha_skip", C.c_float), ("tau_stop", C.c_float), ("depth_key", C.c_int32),
                ("nan_maha", C.c_int32)]


class EgsGaussianParams(C.Structure):
    """Mirror of `struct EgsGaussianParams`: device pointers of the six training tensors (or their moments)."""
    _fields_ = [(k, C.c_void_p) for k in ("pws", "low_shs", "high_shs", "alphas_raw", "scales_raw", "rots_raw")]


class EgsAdamGroup(C.Structure):
    """Mirror of `struct EgsAdamGroup`."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("count", C.c_int64), ("lr", C.c_float), ("step", C.c_int32)]


_P = C.c_void_p
_PP = C.POINTER(EgsPolicy)
_PG = C.POINTER(EgsGaussianParams)
_f = C.c_float
_i = C.c_int
_i64 = C.c_int64
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/egs_hip.h declares
SIGNATURES = {
    "egs_abi_version": (_i, []),
    "egs_last_error_string": (C.c_char_p, []),
    "egs_policy_gsplatcu": (None, [_PP]),
    "egs_policy_forward_cpu": (None, [_PP]),
    "egs_project": (_i, [_i, _P, _P, _P, _f, _f, _f, _f, _PP, _P, _P, _P, _P, _P]),
    "egs_cov3d": (_i, [_i, _P, _P, _P, _PP, _P, _P, _P, _P]),
    "egs_cov2d": (_i, [_i, _P, _P, _P, _P, _f, _f, _f, _f, _PP, _P, _P, _P, _P]),
    "egs_sh2color": (_i, [_i, _i, _P, _P, _P, _P, _P, _P, _P]),
    "egs_inv_cov2d": (_i, [_i, _P, _P, _PP, _P, _P, _P, _P]),
    "egs_splat_bin_ws_bytes": (_sz, [_i]),
    "egs_splat_draw_ws_bytes": (_sz, [_i, _i64, _i, _i]),
    "egs_splat_bin": (_i, [_i, _i, _i, _P, _P, _P, _PP, _i, _P, _sz, _P, _P]),
    "egs_splat_draw": (_i, [_i, _i64, _i, _i, _P, _P, _P, _P, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P]),
    "egs_splat_bin_mb": (_i, [_i, _i, _i, _P, _P, _P, _PP, _i, _P, _sz, _P, _P, _P]),
    "egs_splat_draw_dev": (_i, [_i, _i64, _P, _i, _i, _P, _P, _P, _P, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P]),
    "egs_splat_bwd_ws_bytes": (_sz, [_i]),
    "egs_splat_bwd": (_i, [_i, _i64, _i, _i, _P, _P, _P, _P, _P, _PP, _P, _P, _P, _P, _P, _P, _sz,
                           _P, _P, _P, _P, _P]),
    "egs_pack_records": (_i, [_i, _i, _i, _P, _P, _P, _P, _P, _PP, _P, _P]),
    "egs_splat_bwd_rec": (_i, [_i, _i64, _i, _i, _P, _PP, _P, _P, _P, _P, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P]),
    "egs_splat_bwd_rec_lists": (_i, [_i, _i64, _i, _i, _P, _PP, _P, _P, _P, _P, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _i,
                                     _P]),
    "egs_splat_bin_pack": (_i, [_i, _i, _i, _P, _P, _P, _P, _P, _P, _PP, _i, _P, _sz, _P, _P, _P, _P, _P, _P]),
    "egs_splat_draw_rec_plain": (_i, [_i, _i64, _i, _i, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P]),
    "egs_splat_draw_rec_dev_plain": (_i, [_i, _i64, _P, _i, _i, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P, _P, _i,
                                          _P]),
    "egs_pair_stamp_words": (_sz, [_i]),
    "egs_pack_records_validate": (_i, [_i, _i, _i, _P, _P, _P, _P, _PP, _P, _P, _P, _i64, _P, _P, _P]),
    "egs_strip_list_masks": (_i, [_i64, _P, _P, _P, _P]),
    "egs_sort_pairs_ws_bytes": (_sz, [_i64]),
    "egs_sort_pairs": (_i, [_i64, _P, _P, _P, _P, _i, _i, _P, _sz, C.POINTER(C.c_int), _P]),
    "egs_scan_ws_bytes": (_sz, [_i64]),
    "egs_exclusive_scan_u32": (_i, [_i64, _P, _P, _P, _P, _P, _sz, _P]),
    "egs_words_differ": (_i, [_P, _P, _i64, _P, _P]),
    "egs_chain_rule": (_i, [_i, _i] + [_P] * 16 + [_P]),
    "egs_fused_forward": (_i, [_i, _i] + [_P] * 8 + [_f] * 4 + [_i, _i, _PP] + [_P] * 8 + [_i, _i, _P, _sz, _P, _P, _P]),
    "egs_fused_forward_raw": (_i, [_i, _i] + [_P] * 9 + [_f] * 4 + [_i, _i, _PP] + [_P] * 8
                              + [_i, _i, _P, _sz, _P, _P, _P]),
    "egs_fused_backward_raw": (_i, [_i, _i, _i64, _i, _i] + [_P] * 9 + [_f] * 4 + [_PP] + [_P] * 11 + [_P, _sz]
                               + [_P] * 7 + [_P, _P, _P, _i, _i, _i, _P, _sz, _P]),
    "egs_seg_ws_bytes": (_sz, [_i64, _i, _i]),
    "egs_seg_config": (_i, [_i, _i, C.POINTER(C.c_int)]),
    "egs_splat_draw_rec_seg": (_i, [_i, _i64, _P, _i, _i, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P, _P,
                                    _i, _i, _P, _sz, _P, _P, _P, _P]),
    "egs_seg_rebuild_ws_bytes": (_sz, [_i64, _i, _i]),
    "egs_splat_bwd_seg": (_i, [_i, _i64, _i, _i, _P, _P, _P, _P, _P, _PP, _P, _P, _P, _P, _P, _P, _sz, _P, _P,
                               _P, _P, _P, _P, _i, _P, _sz, _i, _P, _P]),
    "egs_mailbox_peek": (_i, [_P, _i, C.POINTER(C.c_uint32)]),
    "egs_mailbox_clear": (_i, [_P, _i]),
    "egs_sh_grad_views": (_i, [_i, _i, _i, _P, _P, _i64, _f, _P, _P, _i, _P]),
    "egs_tile_order_len": (_sz, [_i, _i]),
    "egs_splat_draw_rec": (_i, [_i, _i64, _i, _i, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _P]),
    "egs_splat_draw_rec_dev": (_i, [_i, _i64, _P, _P, _i, _i, _P, _PP, _P, _P, _sz, _P, _P, _P, _P, _P, _P, _P, _P,
                                    _i, _i, _P]),
    "egs_hbm_copy_probe": (_i, [_P, _P, _sz, _P]),
    "egs_clock_probe": (_i, [_P, _i, _P]),
    "egs_mailbox_create": (_P, [_i]),
    "egs_mailbox_destroy": (None, [_P]),
    "egs_mailbox_post": (_i, [_P, _i, _P, _P]),
    "egs_mailbox_slot": (_P, [_P, _i]),
    "egs_mailbox_arm": (_i, [_P, _i, _P]),
    "egs_mailbox_fetch": (_i, [_P, _i, _i, C.POINTER(C.c_uint32)]),
    "egs_fused_backward_ws_bytes": (_sz, [_i]),
    "egs_fused_backward": (_i, [_i, _i, _i64, _i, _i] + [_P] * 8 + [_f] * 4 + [_PP] + [_P] * 11 + [_P, _sz]
                           + [_P] * 6 + [_P, _P, _P, _i, _i, _i, _P, _sz, _P]),
    "egs_gau_loss_ws_bytes": (_sz, [_i, _i]),
    "egs_gau_loss": (_i, [_i, _i, _P, _P, _f, _f, _P, _sz, _P, _P, _P]),
    "egs_density_accumulate": (_i, [_i, _P, _P, _i, _P, _P, _P]),
    "egs_densify_ws_bytes": (_sz, [_i]),
    "egs_densify_plan": (_i, [_i, _P, _P, _P, _P, _f, _f, _f, _f, _P, _P, _sz, _P, _P]),
    "egs_densify_apply": (_i, [_i, _i, _i, _i, _i, _P, _P, _PG, _PG, _PG, _PG, _PG, _PG, _P, C.c_uint64,
                               C.c_uint64, _P]),
    "egs_reset_alpha": (_i, [_i, _f, _P, _P, _P, _P]),
    "egs_adam_step": (_i, [_i, C.POINTER(EgsAdamGroup), C.c_double, C.c_double, C.c_double, _P]),
    "egs_adam_sh_factored": (_i, [_i, _i, _i, _P, _P, _i64, _f, C.POINTER(EgsAdamGroup), C.POINTER(EgsAdamGroup),
                                  C.c_double, C.c_double, C.c_double, _P]),
    "egs_nn_sqdist_ws_bytes": (_sz, [_i]),
    "egs_nn_sqdist": (_i, [_i, _P, _P, _sz, _P, _P]),
    "egs_viewer_prep": (_i, [_i, _i, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _f, _f, _P, _P, _P]),
    "egs_prof_enable": (_i, [_i]),
    "egs_prof_set_filter": (None, [C.c_char_p]),
    "egs_prof_reset": (None, []),
    "egs_prof_report": (_i, [C.c_char_p, _sz]),
}

_lib = None


class EgsLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libegs_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise EgsLibraryError("building libegs_hip.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


def load():
    """Load the library once; raise EgsLibraryError if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgsLibraryError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C %s`). There is no CPU fallback." % (LIB_PATH, CSRC))
    # torch must own the HIP runtime in this process: libegs_hip.so's DT_NEEDED
    # libamdhip64.so.7 then resolves (by SONAME) to the copy torch already loaded.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EgsLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EgsLibraryError("libegs_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.egs_abi_version() != ABI_VERSION:
        raise EgsLibraryError("libegs_hip.so ABI %d != expected %d" % (lib.egs_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().egs_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(msg)
