"""Loader for librenderbaby_hip.so (the C ABI of include/rb_abi.h).

The product path has no fallback: if the HIP library is missing or fails to
load, importing the engine raises.  ``torch`` is imported first on purpose: its
wheel bundles a HIP runtime with the same SONAME (libamdhip64.so.7), and a
process must not end up with two HIP runtimes if device pointers are to be
shared with torch.distributed (the RCCL gather).
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RB_LIBRARY_PATH: an alternative build of the library (A/B experiments with compile-time knobs)
LIB_PATH = os.environ.get("RB_LIBRARY_PATH") or os.path.join(_HERE, "librenderbaby_hip.so")
_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "There is no CPU fallback for the render path.")
    try:
        import torch  # noqa: F401  (see module docstring)
    except Exception:  # pragma: no cover - torch is optional for pure C-ABI use
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
    sig = {
        "rb_create": (vp, [P(abi.Config)]),
        "rb_create_ex": (vp, [P(abi.Config), P(abi.Options)]),
        "rb_create_multi": (vp, [P(abi.Config), P(abi.Options), P(C.c_int32), u32]),
        "rb_comm_unique_id": (i32, [vp]),
        "rb_comm_available": (i32, []),
        "rb_comm_init_rank": (i32, [vp, vp, u32, u32]),
        "rb_comm_info": (i32, [vp, P(u32), P(u32), P(C.c_float)]),
        "rb_destroy": (None, [vp]),
        "rb_update": (i32, [vp, P(abi.Config)]),
        "rb_render": (i32, [vp, vp]),
        "rb_render_config": (i32, [vp, P(abi.Config), vp]),
        "rb_iter_begin": (i32, [vp, P(abi.Config)]),
        "rb_iter_has_next": (i32, [vp]),
        "rb_iter_next": (i32, [vp, vp]),
        "rb_iter_destroy": (None, [vp]),
        "rb_iter_set_passes_per_frame": (i32, [vp, u32]),
        "rb_last_error": (C.c_char_p, [vp]),
        "rb_get_size": (i32, [vp, P(u32), P(u32)]),
        "rb_clear": (i32, [vp]),
        "rb_dispatch": (i32, [vp, u32, u32]),
        "rb_reserve": (i32, [vp, u32]),
        "rb_sync": (i32, [vp]),
        "rb_read_rgba": (i32, [vp, vp]),
        "rb_read_accumulation": (i32, [vp, vp]),
        "rb_device_rgba": (i32, [vp, P(vp), P(sz)]),
        "rb_host_alloc": (vp, [sz]),
        "rb_host_free": (None, [vp]),
        "rb_local_rows": (i32, [vp, P(u32), P(u32)]),
        "rb_global_row": (i32, [vp, u32, P(u32)]),
        "rb_shard_layout": (i32, [u32, u32, u32, u32, P(u32), P(u32)]),
        "rb_shard_global_row": (u32, [u32, u32, u32, u32]),
        "rb_get_stats": (i32, [vp, P(abi.Stats)]),
        "rb_reset_stats": (i32, [vp]),
        "rb_last_dispatch_ms": (i32, [vp, P(C.c_float)]),
        "rb_bvh_build": (i32, [vp, sz, vp, sz, P(sz), vp]),
        "rb_debug_chunk_tree": (i32, [vp, sz, vp, sz, vp, sz, vp]),
        "rb_measure_l1_gather": (i32, [i32, C.c_uint64, P(C.c_double)]),
        "rb_debug_math": (i32, [vp, vp, vp, u32]),
        "rb_debug_walk_profile": (i32, [vp, i32]),
        "rb_debug_rcp_exhaustive": (i32, [u32, vp]),
        "rb_debug_div_exhaustive": (i32, [u32, u32, u32, u32, u32, u32, vp]),
        "rb_last_kernel_name": (C.c_char_p, [vp]),
        "rb_fast_bvh_builder": (C.c_char_p, [vp, vp]),
        "rb_sphere_tree_builder": (C.c_char_p, [vp, vp]),
        "rb_chunk_tree_builder": (C.c_char_p, [vp, vp]),
        "rb_debug_engine_chunk_tree": (i32, [vp, vp]),
        "rb_version": (C.c_char_p, []),
        "rb_device_name": (i32, [i32, C.c_char_p, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def source_fingerprint():
    """Short hash of the library's sources (csrc/ + the ABI header): profiles are stamped with it so that a
    number measured on another build is never quoted for this one."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(os.path.join(_HERE, "csrc")))
    for name in files:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(os.path.dirname(_HERE), "include", "rb_abi.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


EXPORTS = ["rb_create", "rb_create_ex", "rb_create_multi", "rb_comm_available", "rb_comm_unique_id", "rb_comm_init_rank", "rb_comm_info", "rb_destroy", "rb_update", "rb_render", "rb_render_config",
           "rb_iter_begin", "rb_iter_has_next", "rb_iter_next", "rb_iter_destroy", "rb_iter_set_passes_per_frame", "rb_last_error",
           "rb_get_size", "rb_clear", "rb_dispatch", "rb_reserve", "rb_sync", "rb_read_rgba", "rb_read_accumulation",
           "rb_device_rgba", "rb_host_alloc", "rb_host_free", "rb_local_rows", "rb_global_row", "rb_shard_layout", "rb_shard_global_row", "rb_get_stats", "rb_reset_stats",
           "rb_last_dispatch_ms", "rb_bvh_build", "rb_debug_chunk_tree", "rb_measure_l1_gather", "rb_debug_math", "rb_debug_walk_profile", "rb_debug_rcp_exhaustive", "rb_debug_div_exhaustive", "rb_last_kernel_name", "rb_fast_bvh_builder", "rb_sphere_tree_builder", "rb_chunk_tree_builder", "rb_debug_engine_chunk_tree", "rb_version", "rb_device_name"]
