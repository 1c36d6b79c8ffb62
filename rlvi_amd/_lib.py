"""ctypes binding of librlvi_gfx950.so (the C ABI declared in include/rlvi_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  Nothing here imports oracle/.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RLVI_LIB_PATH: load a differently-built copy of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RLVI_LIB_PATH") or os.path.join(HERE, "librlvi_gfx950.so")

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_int = ctypes.c_int

# name -> (restype, argtypes); mirrors include/rlvi_hip.h one for one
SIGNATURES = {
    "rlvi_abi_version": (_int, []),
    "rlvi_error_string": (ctypes.c_char_p, [_int]),
    "rlvi_tune_set": (_int, [ctypes.c_char_p, _int]),
    "rlvi_tune_unset": (_int, [ctypes.c_char_p]),
    "rlvi_tune_overrides": (_int, [ctypes.c_char_p, _int]),
    "rlvi_device_cus": (_int, []),
    "rlvi_device_pci_bus_id": (_int, [ctypes.c_char_p, _int]),
    "rlvi_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "rlvi_workspace_init": (_int, [_vp, ctypes.c_size_t, _vp]),
    "rlvi_workspace_status": (_int, [_vp, ctypes.POINTER(ctypes.c_int32), _vp]),
    "rlvi_workspace_clear_status": (_int, [_vp, _vp]),
    "rlvi_workspace_set_option": (_int, [_vp, ctypes.c_char_p, _int]),
    "rlvi_workspace_reset_warm": (_int, [_vp, _vp]),
    "rlvi_workspace_last_mstep_form": (_int, [_vp]),
    "rlvi_workspace_region": (ctypes.c_size_t, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]),
    "rlvi_peer_inbox_bytes": (ctypes.c_size_t, []),
    "rlvi_peer_alloc": (_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "rlvi_peer_free": (_int, [_vp]),
    "rlvi_peer_export": (_int, [_vp, ctypes.c_char_p]),
    "rlvi_peer_open": (_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "rlvi_peer_close": (_int, [_vp]),
    "rlvi_peer_can_access": (_int, [_int]),
    "rlvi_workspace_clear_peers": (_int, [_vp, _vp]),
    "rlvi_workspace_set_peers": (_int, [_vp, _int, _int, ctypes.POINTER(ctypes.c_void_p), _vp]),
    "rlvi_threshold_truncate_sharded_f32": (_int, [_vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp]),
    "rlvi_estep_sharded_check": (_int, [_i64, _i64, _int, _int]),
    "rlvi_threshold_sharded_check": (_int, [_i64, _i64]),
    "rlvi_estep_sharded_f32": (_int, [_vp, _vp, _i64, _i64, _f32, _int, _i64, _vp, _vp, _vp, _vp]),
    "rlvi_mstep_fwd_bwd_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32,
                                      _vp, _i64, _vp, _vp, _vp]),
    "rlvi_mstep_fwd_bwd_bf16": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32,
                                       _vp, _i64, _vp, _vp, _vp]),
    "rlvi_mstep_reduce_f32": (_int, [_vp, _f64, _vp, _vp]),
    "rlvi_estep_deep_f32": (_int, [_vp, _vp, _i64, _f32, _int, _vp, _vp, _vp, _vp]),
    "rlvi_epoch_end_f32": (_int, [_vp, _vp, _i64, _f32, _int, _int, _f32, _vp, _i64, _vp, _vp,
                                  _vp, _vp]),
    "rlvi_fn_threshold_f32": (_int, [_vp, _i64, _f32, _vp, _vp, _vp]),
    "rlvi_threshold_truncate_f32": (_int, [_vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp]),
    "rlvi_truncate_f32": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "rlvi_select_smallest_f32": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "rlvi_topk_hits_f32": (_int, [_vp, _i64, _vp, _i64, _i64, ctypes.POINTER(ctypes.c_int32), _int, _vp, _vp]),
    "rlvi_topk_hits_bf16": (_int, [_vp, _i64, _vp, _i64, _i64, ctypes.POINTER(ctypes.c_int32), _int, _vp, _vp]),
    "rlvi_fused_em_f32": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _int,
                                 _vp, _i64, _vp, _vp, _vp, _vp]),
    "rlvi_update_weights_f64": (_int, [_vp, _i64, _f64, _int, _vp, _vp, _vp, _vp]),
    "rlvi_update_weights_online_f64": (_int, [_vp, _i64, _f64, _int, _vp, _vp, _vp, _vp]),
    "rlvi_wls_solve_f64": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "rlvi_linreg_losses_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "rlvi_logistic_nll_f64": (_int, [_vp, _vp, _f64, _i64, _i64, _vp, _vp]),
    "rlvi_linear_regression_check": (_int, [_i64, _i64]),
    "rlvi_linear_regression_f64": (_int, [_vp, _vp, _i64, _i64, _int, _f64, _f64, _int, _vp, _vp, _vp, _vp, _vp]),
    "rlvi_sample_weight_online_f64": (_int, [_vp, _vp, _f64, _int, _i64, _i64, _f64, _int, _vp, _vp, _vp, _vp]),
    "rlvi_stream_copy": (_int, [_vp, _vp, ctypes.c_size_t, _vp]),
}

_lib = None
ABI_VERSION = 3          # RLVI_ABI_VERSION of include/rlvi_hip.h


class RlviError(RuntimeError):
    pass


# device-side status flags (include/rlvi_hip.h)
ST_RANGE, ST_TIMEOUT, ST_NOCONV, ST_SINGULAR = 1, 2, 4, 8
_ST_TEXT = {
    ST_RANGE: "a label or sample index was out of range (the reference raises an IndexError there)",
    ST_TIMEOUT: "an inter-workgroup wait timed out: the cooperating workgroups were not all resident "
                "(another process on this GPU?); pi / threshold were left as they were",
    ST_NOCONV: "the E-step did not reach its fixed point (non-finite residuals?)",
    ST_SINGULAR: "weighted least squares: the Gram matrix is not positive definite",
}


def status_message(status):
    return "; ".join(t for b, t in _ST_TEXT.items() if status & b) or "ok"


def load():
    """Load the shared library and bind every symbol of the header (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RlviError(
            f"{LIB_PATH} is missing: build it with `python -m rlvi_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: librlvi_gfx950.so must bind to the HIP runtime torch has loaded (same
    # SONAME), otherwise the process ends up with two runtimes and ours sees no device
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.rlvi_abi_version() != ABI_VERSION:
        raise RlviError("librlvi_gfx950.so ABI version mismatch")
    _lib = L
    return L


def tune_overrides():
    """Names of the process-wide knobs that carry a rlvi_tune_set value right now."""
    buf = ctypes.create_string_buffer(4096)
    n = load().rlvi_tune_overrides(buf, len(buf))
    return [s for s in buf.value.decode().split(",") if s] if n else []


def check(rc, what):
    if rc != 0:
        msg = load().rlvi_error_string(rc)
        raise RlviError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")
