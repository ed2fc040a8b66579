"""ctypes binding of include/zdr.h (libzdr_hip.so).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
so a GPU box can never silently run anything but the hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libzdr_hip.so")

COLLOCATED, DIRECT, PATH, UVGRAD = 0, 1, 2, 3
SAMPLER_CMJ, SAMPLER_PMJ02BN = 0, 1
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_BVH = 0, 1, 2
ABI_VERSION = 3                # ZDR_ABI_VERSION of the include/zdr.h this binding mirrors
PRB_MODES = {"expectation": 0, "detached": 1, "literal": 2}
INTEGRATORS = {"collocated": COLLOCATED, "direct": DIRECT, "path": PATH}   # render.py:65-69
SAMPLERS = {"cmj": SAMPLER_CMJ, "corrmj": SAMPLER_CMJ, "pmj02bn": SAMPLER_PMJ02BN}
ACCELS = {"auto": ACCEL_AUTO, "brute": ACCEL_BRUTE, "bvh": ACCEL_BVH}
COUNTER_NAMES = ("samples", "closest_rays", "closest_hits", "shadow_rays", "shaded_vertices",
                 "emitter_hits_bsdf", "nan_samples", "shadow_rays_traced")

# every symbol include/zdr.h declares
EXPORTS = ("zdr_version", "zdr_abi_version", "zdr_last_error", "zdr_scene_create", "zdr_scene_destroy", "zdr_scene_info",
           "zdr_scene_set_emissions", "zdr_scene_set_envmap", "zdr_scene_set_pmj02bn_tables", "zdr_render_forward", "zdr_render_backward",
           "zdr_render_stats", "zdr_scene_check", "zdr_trace_closest", "zdr_trace_any", "zdr_sampler_dump", "zdr_vertex_sampler_dump", "zdr_path_dump", "zdr_debug_build_accel", "zdr_debug_never_occluders")


class CameraPOD(C.Structure):
    _fields_ = [("fov", C.c_float), ("origin", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("integrator", C.c_int32), ("sampler", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
        ("spp", C.c_uint32), ("seed", C.c_uint32), ("use_tent", C.c_int32),
        ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
        ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
        ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
        ("camera", CameraPOD), ("tex_h", C.c_int32), ("tex_w", C.c_int32),
        ("tile_shard_index", C.c_int32), ("tile_shard_count", C.c_int32), ("prb_mode", C.c_int32),
    ]


class SceneInfo(C.Structure):
    _fields_ = [("ntris", C.c_uint32), ("nverts", C.c_uint32), ("ninst", C.c_uint32), ("light_count", C.c_uint32),
                ("accel", C.c_int32), ("bvh_nodes", C.c_uint32), ("bvh_max_depth", C.c_uint32), ("bvh_stack_entries", C.c_uint32), ("device", C.c_int32),
                ("device_bytes", C.c_uint64)]


class ZdrError(RuntimeError):
    pass


_LIB = None


def lib():
    """Loads libzdr_hip.so (building it in-tree first if the sources are newer)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    from . import build as _build
    if _build.stale():
        _build.build()
    if not os.path.exists(LIB_PATH):
        raise ZdrError(f"{LIB_PATH} is missing: the zdr HIP back end was not built (python -m zdr_amd.build)")
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p   # raw addresses (host numpy or device data_ptr)
    L.zdr_version.restype = C.c_char_p
    L.zdr_last_error.restype = C.c_char_p
    L.zdr_scene_create.argtypes = [fp, C.c_uint32, ip, C.c_uint32, ip, fp, fp, C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]
    L.zdr_scene_destroy.argtypes = [vp]
    L.zdr_scene_info.argtypes = [vp, C.POINTER(SceneInfo)]
    L.zdr_scene_set_emissions.argtypes = [vp, fp, vp]
    L.zdr_scene_set_envmap.argtypes = [vp, fp, C.c_uint32, C.c_uint32, fp, ip, fp, C.c_uint32, C.c_uint32]
    L.zdr_scene_set_pmj02bn_tables.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    L.zdr_render_forward.argtypes = [vp, C.POINTER(RenderParams), fp, fp, vp]
    L.zdr_render_backward.argtypes = [vp, C.POINTER(RenderParams), fp, fp, fp, vp]
    L.zdr_render_stats.argtypes = [vp, C.POINTER(RenderParams), fp, C.POINTER(C.c_uint64), vp]
    L.zdr_scene_check.argtypes = [vp, vp]
    L.zdr_trace_closest.argtypes = [vp, fp, C.c_uint32, ip, fp, vp]
    L.zdr_trace_any.argtypes = [vp, fp, C.c_uint32, ip, vp]
    L.zdr_sampler_dump.argtypes = [vp, C.c_int32, C.c_uint32, C.c_uint32, ip, C.c_uint32, C.c_int32, C.c_int32, fp, vp]
    L.zdr_vertex_sampler_dump.argtypes = [vp, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, ip, C.c_uint32, C.c_int32, C.c_int32, fp, C.POINTER(C.c_int32), vp]
    L.zdr_path_dump.argtypes = [vp, C.POINTER(RenderParams), fp, fp, ip, C.c_uint32, C.c_int32, fp, vp]
    L.zdr_debug_build_accel.argtypes = [fp, C.c_uint32, C.c_int, fp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), ip, fp]
    L.zdr_debug_never_occluders.argtypes = [fp, C.c_uint32, vp, vp]
    for name in EXPORTS:
        getattr(L, name)          # AttributeError here = header and library disagree
    if L.zdr_abi_version() != ABI_VERSION:
        raise ZdrError(f"{LIB_PATH} speaks ABI {L.zdr_abi_version()}, this binding ABI {ABI_VERSION}: rebuild (python -m zdr_amd.build --force)")
    _LIB = L
    return L


def check(rc: int):
    if rc != 0:
        raise ZdrError(f"libzdr_hip error {rc}: {lib().zdr_last_error().decode()}")
