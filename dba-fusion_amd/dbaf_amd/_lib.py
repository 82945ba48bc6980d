"""Loader for the gfx950 C-ABI library (dba-fusion_amd/lib/libdba_hip.so, include/dba_hip.h).

The library is built in-tree by `make lib` / `__graft_entry__.build()`.  There is NO fallback: if the
shared object is missing or a call fails, the product path raises -- it never computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBA_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libdba_hip.so")

DBA_F32, DBA_F16, DBA_F64 = 0, 1, 2
_ERR = {-1: "DBA_ERR_ARG", -2: "DBA_ERR_WORKSPACE", -3: "DBA_ERR_HIP", -4: "DBA_ERR_UNSUPPORTED"}

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p


class BaLayout(ctypes.Structure):
    _fields_ = [(n, c_size_t) for n in ("H", "b", "dx", "meta", "E", "Q", "w", "kx")] + \
               [("P", c_int), ("Mmax", c_int), ("nchunks", c_int)]


class ShardExchange(ctypes.Structure):
    """dba_shard_exchange of include/dba_hip.h"""
    _fields_ = [("world", c_int), ("rank", c_int), ("comm", c_void_p), ("peer_regions", c_void_p),
                ("peer_epoch", ctypes.POINTER(ctypes.c_uint)), ("peer_max_doubles", c_size_t), ("peer_status", c_void_p),
                ("band_idx", c_void_p), ("band_len", c_size_t), ("band_buf", c_void_p),
                ("my_rows", c_void_p), ("n_mine", c_int), ("kmax", c_int), ("all_rows", c_void_p), ("all_slots", c_void_p),
                ("n_all", c_int), ("send", c_void_p), ("recv", c_void_p)]


# every exported symbol of include/dba_hip.h with its (restype, argtypes); pointers are void*
_P = c_void_p
SYMBOLS = {
    "dba_version": (ctypes.c_char_p, []),
    "dba_last_error": (ctypes.c_char_p, []),
    "dba_ba_workspace_bytes": (c_size_t, [c_int] * 6),
    "dba_ba_get_layout": (c_int, [c_int] * 6 + [ctypes.POINTER(BaLayout)]),
    "dba_ba_prepare": (c_int, [_P, _P] + [c_int] * 6 + [_P, c_size_t, _P]),
    "dba_ba_prepare_keyed": (c_int, [_P, _P] + [c_int] * 8 + [_P, c_size_t, _P]),
    "dba_ba_workspace_init": (c_int, [c_int] * 6 + [_P, c_size_t, _P]),
    "dba_ba_solver_verdict": (c_int, [c_int] * 6 + [_P, c_size_t]),
    "dba_ba_poll_eta_error": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dba_ba_poll_eta_error_ws": (c_int, [c_int] * 6 + [_P, c_size_t, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dba_ba_gather_edges": (c_int, [_P] * 4 + [c_int, _P, c_int] + [_P] * 4 + [c_int] * 3 + [_P] * 5),
    "dba_ba_linearize": (c_int, [_P] * 7 + [c_int] + [_P] * 3 + [c_int] * 6 + [c_float, _P, c_size_t, _P]),
    "dba_ba_reduce": (c_int, [_P] * 3 + [c_int] * 7 + [_P, c_size_t, _P]),
    "dba_ba_schur_select": (c_int, [c_int]),
    "dba_ba_schur_select_thread": (c_int, [c_int]),
    "dba_ba_schur_auto_form": (c_int, [c_int, c_int]),
    "dba_ba_schur_thread_form": (c_int, []),
    "dba_ba_schur_generation": (c_int, []),
    "dba_ba_set_deterministic": (c_int, [c_int]),
    "dba_ba_set_solve_check": (c_int, [c_int]),
    "dba_ba_solve_check": (c_int, [c_int] * 6 + [c_float, c_float, _P, c_size_t, _P]),
    "dba_ba_symmetrize": (c_int, [c_int] * 6 + [_P, c_size_t, _P]),
    "dba_ba_solve": (c_int, [c_int] * 6 + [c_float, c_float, _P, c_size_t, _P]),
    "dba_ba_solve_skyline": (c_int, [c_int] * 6 + [c_float, c_float, _P, _P, c_size_t, _P]),
    "dba_ba_update": (c_int, [_P] * 5 + [c_int] * 8 + [_P, _P, c_size_t, _P]),
    "dba_ba_shard_front": (c_int, [_P] * 7 + [c_int] + [_P] * 3 + [c_int] * 6 + [c_float, c_int, _P, c_size_t, _P]),
    "dba_ba_shard_back": (c_int, [_P] * 5 + [c_int] * 6 + [c_float, c_float, c_int, _P, c_int, _P, c_size_t, _P]),
    "dba_comm_unique_id": (c_int, [_P]),
    "dba_comm_create": (c_int, [_P, c_int, c_int, ctypes.POINTER(_P)]),
    "dba_comm_destroy": (c_int, [_P]),
    "dba_comm_info": (c_int, [_P, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dba_comm_allreduce_f64": (c_int, [_P, c_size_t, _P]),
    "dba_ba_sharded_run": (c_int, [_P] * 7 + [c_int] + [_P] * 3 + [c_int] * 7 + [c_float, c_float, c_float, c_int, _P, c_int,
                                                                                  c_int, ctypes.POINTER(ShardExchange), _P,
                                                                                  c_size_t, _P]),
    "dba_ba": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 7 + [c_float, c_float, c_int, _P, _P, _P,
                                                                     c_size_t, _P]),
    "dba_ba_prepared": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 7 + [c_float, c_float, c_int, _P, _P, _P,
                                                                              c_size_t, _P, c_int]),
    "dba_ba_run": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 7 + [c_float, c_float, c_int, _P, _P, _P,
                                                                         c_size_t, _P, c_int, c_int, c_float]),
    "dba_bacore_hessian": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 6 + [_P, _P, _P, c_size_t, _P]),
    "dba_bacore_hessian_run": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 6 + [_P, _P, _P, c_size_t, _P, c_int]),
    "dba_bacore_staging": (c_int, [c_int] * 6 + [_P, c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "dba_bacore_export_host": (c_int, [c_int] * 6 + [_P, c_size_t, _P, c_int, _P, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]),
    "dba_bacore_hessian_host": (c_int, [_P] * 7 + [c_int] + [_P, _P] + [c_int] * 6 + [_P, c_size_t, _P, c_int, c_int, _P,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]),
    "dba_bacore_retract": (c_int, [_P] * 4 + [c_int] * 6 + [_P, _P, _P, _P, c_size_t, _P]),
    "dba_bacore_optimize": (c_int, [_P, _P] + [c_int] * 6 + [c_float, c_float, _P, _P, c_size_t, _P]),
    "dba_corr_index_forward": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "dba_corr_lookup_pyramid": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "dba_corr_volume_build_sheared_supported": (c_int, [c_int] * 6),
    "dba_corr_volume_build_sheared": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P, c_size_t, _P]),
    "dba_corr_sheared_plane_elems": (c_int, [c_int, c_int]),
    "dba_corr_sheared_tiled": (c_int, [c_int, c_int]),
    "dba_corr_sheared_grid": (c_int, [c_int, c_int, _P, _P]),
    "dba_corr_lookup_select": (c_int, [c_int]),
    "dba_corr_lookup_arm_timing": (c_int, [_P, _P]),
    "dba_corr_shear_level": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "dba_corr_lookup_pyramid_sheared": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "dba_corr_lookup_level_sheared": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "dba_corr_lookup_level_sheared_slots": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "dba_corr_shear_level_slots": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "dba_corr_volume_build_sheared_slots": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P, c_size_t, _P]),
    "dba_corr_lookup_pyramid_sheared_slots": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "dba_corr_lookup_pyramid_slots": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "dba_corr_lookup_reproject_sheared": (c_int, [_P] * 10 + [c_int] * 7 + [_P]),
    "dba_corr_index_backward": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "dba_corr_volume_scratch_bytes": (c_size_t, [c_int] * 6),
    "dba_corr_once_pyramid_bytes": (c_size_t, [c_int] * 6),
    "dba_corr_build_lookup_once_sheared": (c_int, [_P, _P, _P, _P, _P, c_size_t, _P, c_size_t] + [c_int] * 8 + [_P]),
    "dba_corr_volume_build": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P, c_size_t, _P]),
    "dba_altcorr_forward": (c_int, [_P] * 4 + [c_int] * 8 + [_P]),
    "dba_altcorr_forward_t": (c_int, [_P] * 4 + [c_int] * 9 + [_P]),
    "dba_altcorr_pyramid_forward": (c_int, [_P] * 6 + [c_int] * 8 + [_P]),
    "dba_altcorr_pyramid_forward_f16maps": (c_int, [_P] * 6 + [c_int] * 7 + [_P]),
    "dba_altcorr_backward": (c_int, [_P] * 6 + [c_int] * 8 + [_P]),
    "dba_reproject": (c_int, [_P] * 5 + [c_int] * 3 + [_P, _P, _P]),
    "dba_frame_distance": (c_int, [_P] * 5 + [c_int] * 3 + [c_float, _P, _P]),
    "dba_projmap": (c_int, [_P] * 5 + [c_int] * 3 + [_P, _P, _P]),
    "dba_iproj": (c_int, [_P] * 3 + [c_int] * 3 + [_P, _P]),
    "dba_depth_filter": (c_int, [_P] * 5 + [c_int] * 4 + [_P, _P]),
    "dba_peer_exchange_bytes": (c_size_t, [c_size_t]),
    "dba_peer_exchange_create": (c_int, [c_size_t, ctypes.POINTER(_P), _P]),
    "dba_peer_exchange_open": (c_int, [_P, ctypes.POINTER(_P)]),
    "dba_peer_exchange_close": (c_int, [_P, c_int]),
    "dba_peer_allreduce_f64": (c_int, [_P, c_size_t, _P, c_int, c_int, ctypes.c_uint, c_size_t, _P, _P]),
}

_lib = None


def load():
    """dlopen the HIP library and type every entry point. Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "dba_hip: %s not found -- build it with `make lib` (hipcc --offload-arch=gfx950); "
                "there is no CPU fallback for the DBA hot path" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def schur_generation():
    """what decides which Schur form (and therefore which stage-0 tables) is in force besides the graph itself: the
    process-wide selection's generation and the calling thread's pin"""
    lib = load()
    return (lib.dba_ba_schur_generation(), lib.dba_ba_schur_thread_form())


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, str(rc))
        detail = load().dba_last_error().decode() if rc == -3 else ""
        raise RuntimeError("dba_hip: %s failed with %s %s" % (what, msg, detail))
