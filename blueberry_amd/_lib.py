"""ctypes binding of libblueberry_hip.so (C-ABI: include/blueberry_hip.h).

There is deliberately NO fallback here: if the shared library is missing, or
there is no usable MI355X, every compute call raises.  The CPU oracle under
oracle/ is test infrastructure and is never imported from this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BB_LIB: load another build of the same library (kernel-tuning experiments only)
LIB_PATH = os.environ.get("BB_LIB") or os.path.join(_HERE, "libblueberry_hip.so")

BB_OK, BB_ERR_INVALID, BB_ERR_HIP, BB_ERR_STATE, BB_ERR_NOMEM = 0, 1, 2, 3, 4
BB_F32, BB_F64 = 0, 1
BB_KIND_WISH, BB_KIND_COUNTS = 0, 1
BB_PEER_HANDLE_BYTES = 128

c_i32, c_i64, c_dbl = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
p_i32, p_i64, p_dbl = (ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl))
c_int, c_void_p = ctypes.c_int, ctypes.c_void_p


class LayoutInfo(ctypes.Structure):
    """struct bb_layout_info (include/blueberry_hip.h)."""
    _fields_ = [(name, c_i64) for name in (
        "n_bins", "n_pad", "vw", "rows_per_unit", "units_per_tile", "n_blocks", "n_tiles",
        "n_units")]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


# name -> (restype, argtypes); every symbol include/blueberry_hip.h declares
SIGNATURES = {
    "bb_version": (c_int, []),
    "bb_last_error": (ctypes.c_char_p, []),
    "bb_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "bb_band_count": (c_int, [p_dbl, c_i64, c_i32, c_i32, c_int, p_i64]),
    "bb_band_count_rows": (c_int, [p_dbl, c_i64, c_i32, c_i32, c_i64, c_i64, c_int, p_i64]),
    "bb_layout_dense_info": (c_int, [c_i64, c_int, ctypes.POINTER(LayoutInfo)]),
    "bb_layout_dense_tiles": (c_int, [c_i64, c_int, p_i32, p_i32, c_i64]),
    "bb_layout_rank_units": (c_int, [c_i64, c_int, c_int, p_i64, p_i64]),
    "bb_solver_create": (c_int, [ctypes.POINTER(c_void_p), c_i64, c_int, c_int, c_int, c_int,
                                 p_i32, p_i32, c_i64]),
    "bb_solver_destroy": (c_int, [c_void_p]),
    "bb_solver_set_stream": (c_int, [c_void_p, c_void_p]),
    "bb_solver_layout": (c_int, [c_void_p, ctypes.POINTER(LayoutInfo), p_i64, p_i64]),
    "bb_solver_set_wish_dense": (c_int, [c_void_p, p_dbl, c_i64, c_int, c_dbl]),
    "bb_solver_set_wish_sparse": (c_int, [c_void_p, p_i64, p_i64, p_dbl, c_i64, c_int, c_dbl,
                                          p_dbl, p_dbl]),
    "bb_triples_create": (c_int, [ctypes.POINTER(c_void_p), p_dbl, c_i64, c_i32, c_i32, c_int]),
    "bb_triples_destroy": (c_int, [c_void_p]),
    "bb_triples_tiles": (c_int, [c_void_p, c_i64, c_int, ctypes.POINTER(ctypes.c_uint8), c_i64]),
    "bb_solver_set_wish_triples": (c_int, [c_void_p, c_void_p, c_int, c_dbl, p_dbl, p_dbl]),
    "bb_solver_set_wish_from_coords": (c_int, [c_void_p, p_dbl]),
    "bb_solver_set_coords": (c_int, [c_void_p, p_dbl]),
    "bb_solver_get_coords": (c_int, [c_void_p, p_dbl]),
    "bb_solver_set_momentum": (c_int, [c_void_p, c_dbl]),
    "bb_solver_iterate": (c_int, [c_void_p, c_i64, c_dbl]),
    "bb_solver_grad": (c_int, [c_void_p]),
    "bb_solver_apply": (c_int, [c_void_p, c_dbl]),
    "bb_comm_unique_id": (c_int, [c_void_p]),
    "bb_solver_comm_init": (c_int, [c_void_p, c_void_p]),
    "bb_solver_allreduce": (c_int, [c_void_p]),
    "bb_solver_iterate_dist": (c_int, [c_void_p, c_i64, c_dbl]),
    "bb_solver_peer_export": (c_int, [c_void_p, c_void_p]),
    "bb_solver_peer_connect": (c_int, [c_void_p, c_void_p]),
    "bb_solver_iterate_peer": (c_int, [c_void_p, c_i64, c_dbl]),
    "bb_solver_peer_status": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "bb_solver_peer_set_timeout": (c_int, [c_void_p, c_i64]),
    "bb_solver_peer_form": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "bb_solver_peer_set_form": (c_int, [c_void_p, c_int]),
    "bb_solver_comm_world": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "bb_solver_comm_abort": (c_int, [c_void_p]),
    "bb_comm_cached": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    "bb_comm_cached_generation": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int),
                                          ctypes.POINTER(ctypes.c_uint64)]),
    "bb_solver_comm_attach": (c_int, [c_void_p]),
    "bb_solver_comm_detach": (c_int, [c_void_p]),
    "bb_comm_cache_clear": (c_int, []),
    "bb_solver_sync_timeout": (c_int, [c_void_p, c_i64]),
    "bb_solver_exchange_size": (c_int, [c_void_p, p_i64]),
    "bb_solver_get_exchange_buffer": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "bb_solver_set_exchange_buffer": (c_int, [c_void_p, c_void_p]),
    "bb_solver_read_exchange": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_solver_write_exchange": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_solver_matvec_sq": (c_int, [c_void_p, p_dbl, p_dbl]),
    "bb_solver_spectral_init": (c_int, [c_void_p, c_int, p_dbl]),
    "bb_solver_spectral_init_tol": (c_int, [c_void_p, c_int, c_dbl, p_dbl, ctypes.POINTER(c_int), p_dbl]),
    "bb_solver_stress": (c_int, [c_void_p, p_dbl]),
    "bb_solver_get_stress_history": (c_int, [c_void_p, p_dbl, c_i64, p_i64]),
    "bb_solver_sync": (c_int, [c_void_p]),
    "bb_solver_set_timing": (c_int, [c_void_p, c_int]),
    "bb_solver_get_timing": (c_int, [c_void_p, p_dbl, p_dbl, p_i64]),
    "bb_solver_get_step_timing": (c_int, [c_void_p, p_dbl]),
    "bb_solver_measure_event_gap": (c_int, [c_void_p, c_int, p_dbl]),
    "bb_solver_measure_stream_read": (c_int, [c_void_p, c_int, p_dbl]),
    "bb_solver_iteration_path": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "bb_solver_traffic": (c_int, [c_void_p, p_i64, p_i64]),
    "bb_cm_create": (c_int, [ctypes.POINTER(c_void_p), c_i64, c_int]),
    "bb_cm_destroy": (c_int, [c_void_p]),
    "bb_cm_dim": (c_int, [c_void_p, p_i64]),
    "bb_cm_device_ptr": (c_int, [c_void_p, ctypes.POINTER(c_void_p), p_i64, ctypes.POINTER(c_int)]),
    "bb_cm_upload": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_cm_download": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_cm_scatter": (c_int, [c_void_p, p_dbl, c_i64, c_i32]),
    "bb_cm_scatter_ex": (c_int, [c_void_p, p_dbl, c_i64, c_i32, c_i32, ctypes.POINTER(ctypes.c_uint8),
                                 ctypes.POINTER(c_i32)]),
    "bb_cm_normalize": (c_int, [c_void_p, c_i64, p_dbl, p_dbl]),
    "bb_cm_marginals": (c_int, [c_void_p, p_dbl]),
    "bb_cm_filter": (c_int, [c_void_p, c_dbl, p_i64, ctypes.POINTER(ctypes.c_uint8)]),
    "bb_cm_symv": (c_int, [c_void_p, p_dbl, p_dbl]),
    "bb_cm_eigenvector": (c_int, [c_void_p, p_dbl, p_dbl, c_dbl, c_i64, p_i64, p_dbl]),
    "bb_cm_correlation": (c_int, [c_void_p, p_dbl]),
    "bb_cm_release_scratch": (c_int, [c_int]),
    "bb_solver_set_wish_from_cm": (c_int, [c_void_p, c_void_p, c_int, c_dbl]),
    "bb_solver_set_wish_from_cm_block": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_dbl]),
    "bb_solver_set_maps": (c_int, [c_void_p, c_int, p_i64, p_dbl]),
    "bb_solver_set_wish_dense_block": (c_int, [c_void_p, p_dbl, c_i64, c_i64, c_i64, c_int, c_dbl]),
    "bb_solver_stress_maps": (c_int, [c_void_p, p_dbl, c_int]),
    "bb_solver_set_block_steps": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_solver_set_bin_steps": (c_int, [c_void_p, p_dbl, c_i64]),
    "bb_solver_degrees": (c_int, [c_void_p, p_i64, c_i64]),
    "bb_contactmap_scatter": (c_int, [p_dbl, c_i64, c_i32, p_dbl, c_i64, c_int]),
    "bb_contactmap_normalize": (c_int, [p_dbl, c_i64, p_dbl, p_dbl, c_int]),
    "bb_benjamini_hochberg": (c_int, [p_dbl, c_i64, c_i64, p_dbl, c_int]),
    "bb_downsample": (c_int, [ctypes.POINTER(ctypes.c_float), c_i64,
                              ctypes.POINTER(ctypes.c_float), c_i64, c_int]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "blueberry_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as "
            "g; g.build()'` (or ./build.sh) at the repository root. There is no CPU fallback."
            % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def hip_runtimes_loaded():
    """Paths of the libamdhip64 images mapped into this process.  More than one
    means torch's bundled HIP runtime and the system one are both live, and
    stream / event handles cannot cross between them (see solver.exchange_tensor)."""
    paths = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def last_error():
    msg = load().bb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    """Turn a BB_ERR_* status into the Python exception the host API documents."""
    if rc == BB_OK:
        return
    msg = last_error() or ("%s failed (status %d)" % (what, rc))
    if rc == BB_ERR_INVALID:
        raise ValueError(msg)
    if rc == BB_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def as_f64_ptr(arr):
    return arr.ctypes.data_as(p_dbl)
