"""ctypes binding of libskelsplat_hip.so (include/skelsplat_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  Loading fails loudly if the shared object is
missing and cannot be built, and every op raises if it is handed non-ROCm tensors.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libskelsplat_hip.so")
_lib = None

SKS_MAX_VIEWS = 64
SKS_MAX_CHANNELS = 32
SKS_SMALL_P = 256
SKS_ANTIALIASING = 1
SKS_CLAMP01 = 2
SKS_FORCE_BINNED = 4
SKS_DEBUG_SYNC = 8
SKS_NO_NT_STORES = 16
SKS_RAW_PARAMS = 32
SKS_BIN_CLEAN = 64
SKS_RAW_GRADS = 128
SKS_FB_NO_JOIN = 1
SKS_BIN_GROUPS_SHIFT = 16
SKS_SSIM_SCRATCH_BYTES = 64 * 8


def SKS_BIN_GROUPS(n):
    """Flag bits for `n` view groups on the binned path (include/skelsplat_hip.h)."""
    return ((int(n) - 1) & 7) << SKS_BIN_GROUPS_SHIFT


_vp, _i, _u, _f, _sz = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_size_t

# symbol -> (restype, argtypes); mirrors include/skelsplat_hip.h (tests check every declared symbol is exported)
SIGNATURES = {
    "sks_last_error": (C.c_char_p, []),
    "sks_version": (_i, []),
    "sks_scratch_bytes": (_i, [_i, _i, _i, _i, _i, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "sks_forward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u,
                         _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sks_backward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u,
                          _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_forward_backward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u]),
    "sks_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "sks_mean_views": (_i, [_i, _i, _vp, _i, _vp, _vp]),
    "sks_export_lists": (_i, [_i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "sks_masked_l2": (_i, [_i, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sks_masked_l2_loss": (_i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sks_fused_ssim_fwd": (_i, [_i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_fused_ssim_bwd": (_i, [_i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_fused_ssim_bwd_uniform": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "sks_fused_ssim_mean": (_i, [_i, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_knn3_meandist2": (_i, [_i, _vp, _vp, _vp]),
    "sks_knn3_scratch_bytes": (_sz, [_i]),
    "sks_knn3_meandist2_grid": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "sks_heatmaps": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_heatmap_factors": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "sks_heatmap_totals": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_gt_tile_stats": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sks_geometry": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u, _vp, _vp, _vp, _i, _vp]),
    "sks_backward_fused_loss": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _u,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_loop_pack_grads": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sks_loop_adam_step": (_i, [_i, _i, _vp, _vp, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                _f, _vp, _i, _vp]),
    "sks_loop_adam_step_es": (_i, [_i, _i, _vp, _vp, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                   _f, _vp, _i, _vp, _vp, _i, _f, _vp, _vp]),
    "sks_loop_shard_floats": (_sz, [_i, _i, _i]),
    "sks_adam_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "sks_loop_fused_step": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp,
                                 _i, _vp, _vp]),
    "sks_prof_enable": (_i, [_i]),
    "sks_prof_spin": (_i, [C.c_double, _vp]),
    "sks_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "sks_prof_count": (_i, [_i, C.POINTER(C.c_longlong)]),
    "sks_prof_read_quantiles": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
}


def load(build_if_missing=True):
    """Returns the loaded CDLL; raises ImportError with the build error if it cannot be produced."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing:
        try:
            from . import build as _b
            if _b._stale():
                _b.build()
        except Exception as e:  # no hipcc on this box: fall through to the prebuilt .so
            if not os.path.exists(LIB_PATH):
                raise ImportError(f"libskelsplat_hip.so is missing and could not be built: {e}") from e
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found; run `python -m skelsplat_amd.build`")
    lib = C.CDLL(os.environ.get("SKS_LIB_OVERRIDE") or LIB_PATH)   # override: A/B of build variants (tools/ab_libs.sh)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().sks_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def scratch_bytes(V, P, C_, W, H, bin_capacity=0):
    g, b, a = _sz(0), _sz(0), _sz(0)
    check(load().sks_scratch_bytes(V, P, C_, W, H, bin_capacity, C.byref(g), C.byref(b), C.byref(a)), "sks_scratch_bytes")
    return g.value, b.value, a.value


def ptr(t):
    """Device pointer of a tensor, or None.  Empty tensors are the reference's "not provided" sentinel."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def farray(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def prof_count(kind):
    """Bracketed launches of one kind collected since the last read."""
    n = C.c_longlong(0)
    check(load().sks_prof_count(kind, C.byref(n)), "sks_prof_count")
    return n.value


def prof_enable(on, every=1, kinds=(0, 1), recorded=False):
    """Bracket the forward (kind 0) / backward (kind 1) compositor launches with hipEvents: every `every`-th launch of each
    kind in `kinds`.  What was collected so far stays until it is read (a caller may change the stride mid-collection)."""
    skip = sum(1 << (16 + k) for k in (0, 1) if k not in kinds)
    check(load().sks_prof_enable((min(0xffff, max(1, int(every))) | skip | ((1 << 18) if recorded else 0)) if on else 0),
          "sks_prof_enable")


def prof_read_quantiles(kind):
    """(total_ms, launches, (p10_ms, p50_ms, p90_ms)) of the bracketed launches of one kind since the last read."""
    q = (C.c_double * 3)()
    ms, n = C.c_double(0), C.c_longlong(0)
    check(load().sks_prof_read_quantiles(kind, q, C.byref(ms), C.byref(n)), "sks_prof_read_quantiles")
    return ms.value, n.value, (q[0], q[1], q[2])


def prof_read(kind):
    """(total_ms, launches) of the forward (kind 0) / backward (kind 1) compositor kernel since the last read."""
    ms, n = C.c_double(0), C.c_longlong(0)
    check(load().sks_prof_read(kind, C.byref(ms), C.byref(n)), "sks_prof_read")
    return ms.value, n.value
