"""ctypes binding of libbhgpu.so -- one entry per declaration in include/bhgpu.h."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BHGPU_LIB: an A/B build variant (scripts/build_variants.sh), for the scripts/ A/B drivers only.  It must come
# with BHGPU_LIB_OPT_IN=1: a test or bench run that merely inherited the variable would otherwise measure a
# variant while reporting as the product -- load() refuses that.
PRODUCT_LIB = os.path.join(HERE, "libbhgpu.so")
LIB_PATH = os.environ.get("BHGPU_LIB") or PRODUCT_LIB

ABI_VERSION = 4


class bh_config(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("theta", C.c_double),
        ("G", C.c_double),
        ("dt", C.c_double),
        ("max_depth", C.c_int32),
        ("precision", C.c_int32),
        ("reference_compat", C.c_int32),
        ("device", C.c_int32),
        ("n_threads", C.c_int32),
        ("flags", C.c_uint32),
        ("node_capacity", C.c_int64),
    ]


class bh_tree_node(C.Structure):
    _fields_ = [
        ("child", C.c_double * 4),
        ("comx", C.c_double), ("comy", C.c_double), ("mass", C.c_double),
        ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
        ("particle", C.c_double),
    ]


class bh_stats_t(C.Structure):
    _fields_ = [
        ("n_bodies", C.c_int64), ("n_nodes", C.c_int64), ("n_internal", C.c_int64),
        ("steps_done", C.c_int64), ("visits", C.c_uint64), ("interactions", C.c_uint64),
        ("wave_nodes", C.c_uint64),
        ("last_step_ms", C.c_double), ("build_ms", C.c_double), ("walk_ms", C.c_double),
        ("device_bytes", C.c_uint64),
        ("keys_ms", C.c_double), ("sort_ms", C.c_double), ("scan_ms", C.c_double), ("nodes_ms", C.c_double),
        ("build_bytes", C.c_uint64), ("walk_bytes", C.c_uint64), ("wave_quads", C.c_uint64),
        ("sort_spill_buckets", C.c_uint64),
        ("let_tree_ms", C.c_double), ("let_pack_ms", C.c_double),
        ("sort_rerun_buckets", C.c_uint64),
        ("wave_accepts", C.c_uint64), ("walk_launches", C.c_uint64),
    ]


ORB_BINS = 4096
ORB_MAX_CUTS = 63


class bh_orb_cuts(C.Structure):
    _fields_ = [
        ("world", C.c_int32), ("n_cuts", C.c_int32),
        ("box", C.c_double * 4),
        ("axis", C.c_int32 * ORB_MAX_CUTS), ("pad", C.c_int32),
        ("value", C.c_double * ORB_MAX_CUTS),
    ]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_ctx = C.c_void_p

# name -> (restype, argtypes): every symbol include/bhgpu.h declares
SIGNATURES = {
    "bh_abi_version": (C.c_int, []),
    "bh_create": (C.c_int, [C.POINTER(bh_config), C.POINTER(_ctx)]),
    "bh_destroy": (None, [_ctx]),
    "bh_last_error": (C.c_char_p, [_ctx]),
    "bh_upload": (C.c_int, [_ctx, _dp, _dp, _dp, C.c_int64]),
    "bh_download": (C.c_int, [_ctx, _dp, _dp]),
    "bh_initialize": (C.c_int, [_ctx, C.c_int64, C.c_uint64, C.c_int32] + [C.c_double] * 6),
    "bh_download_masses": (C.c_int, [_ctx, _dp]),
    "bh_step": (C.c_int, [_ctx, C.c_int32]),
    "bh_sync": (C.c_int, [_ctx]),
    "bh_build_tree": (C.c_int, [_ctx]),
    "bh_compute_forces": (C.c_int, [_ctx]),
    "bh_get_forces": (C.c_int, [_ctx, _dp]),
    "bh_get_accel": (C.c_int, [_ctx, _dp]),
    "bh_get_interaction_counts": (C.c_int, [_ctx, C.POINTER(C.c_uint32)]),
    "bh_build_info": (C.c_char_p, []),
    "bh_step_times": (C.c_int, [_ctx, _dp, _dp, C.c_int32, C.POINTER(C.c_int32)]),
    "bh_export_tree": (C.c_int, [_ctx, _vp, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]),
    "bh_write_quadtree_file": (C.c_int, [_ctx, C.c_char_p]),
    "bh_stats": (C.c_int, [_ctx, C.POINTER(bh_stats_t)]),
    "bh_set_owned_fraction": (C.c_int, [_ctx, C.c_int32, C.c_int32]),
    "bh_device_state": (C.c_int, [_ctx, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "bh_owned_range": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bh_step_local": (C.c_int, [_ctx]),
    "bh_device_sorted": (C.c_int, [_ctx, C.POINTER(_vp)]),
    "bh_scatter_sorted": (C.c_int, [_ctx]),
    "bh_set_stream": (C.c_int, [_ctx, _vp]),
    "bh_let_local_quads": (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    "bh_let_configure": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    "bh_let_bounds": (C.c_int, [_ctx]),
    "bh_let_pointers": (C.c_int, [_ctx, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "bh_let_build": (C.c_int, [_ctx]),
    "bh_let_walk": (C.c_int, [_ctx]),
    "bh_let_forces": (C.c_int, [_ctx]),
    "bh_let_walk_local": (C.c_int, [_ctx]),
    "bh_let_walk_remote": (C.c_int, [_ctx, C.c_int32]),
    "bh_let_counts": (C.c_int, [_ctx, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "bh_set_ids": (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    "bh_get_ids": (C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    "bh_orb_histogram": (C.c_int, [_ctx, C.POINTER(bh_orb_cuts), C.c_int32, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "bh_migrate_pack": (C.c_int, [_ctx, C.POINTER(bh_orb_cuts), C.POINTER(C.c_int64)]),
    "bh_migrate_pointers": (C.c_int, [_ctx, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "bh_migrate_unpack": (C.c_int, [_ctx, C.c_int64]),
}

_lib = None


def load() -> C.CDLL:
    """Load libbhgpu.so; raise (never fall back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if LIB_PATH != PRODUCT_LIB and os.environ.get("BHGPU_LIB_OPT_IN") != "1":
            raise ImportError(f"BHGPU_LIB={LIB_PATH} selects a build variant, not the product library; set "
                              "BHGPU_LIB_OPT_IN=1 to do that on purpose (scripts/ A/B drivers), or unset BHGPU_LIB")
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m gpu_nbody_simulation_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.bh_abi_version() != ABI_VERSION:
            raise ImportError(f"libbhgpu.so ABI {lib.bh_abi_version()} != binding {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def is_product_library() -> bool:
    """False when an A/B variant was loaded on purpose (BHGPU_LIB + BHGPU_LIB_OPT_IN=1)."""
    return os.path.abspath(LIB_PATH) == os.path.abspath(PRODUCT_LIB)
