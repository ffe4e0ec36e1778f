"""ctypes binding of libpegasus_raster.so (include/pegasus_raster.h).

There is NO fallback: if the library is missing or fails to load, every rasterizer entry point
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# PGR_LIB: an alternative build of the library (stats / timing / variant builds of the measurement scripts load theirs
# from build_variants/ instead of overwriting the product .so)
LIB_PATH = Path(os.environ.get("PGR_LIB") or Path(__file__).resolve().parent / "csrc" / "libpegasus_raster.so")

PGR_ABI_VERSION = 3          # include/pegasus_raster.h PGR_ABI_VERSION
PGR_OK = 0
PGR_ERR_INVALID_ARGUMENT = -1
PGR_ERR_WORKSPACE_TOO_SMALL = -2
PGR_ERR_INSTANCE_OVERFLOW = -3
PGR_ERR_LAUNCH_FAILURE = -4
PGR_NUM_STAGES = 5
STAGE_NAMES = ("preprocess", "bin_count", "bin_scatter", "tile_sort", "composite")


class PgrScene(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("means3d", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3d_precomp", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("sh_degree", C.c_int32), ("sh_stride", C.c_int32), ("scale_modifier", C.c_float),
        ("tie_index", C.c_void_p), ("tie_inv", C.c_void_p), ("shs_rest", C.c_void_p),
    ]


class PgrCamera(C.Structure):
    _fields_ = [
        ("image_width", C.c_int32), ("image_height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("bg", C.c_void_p),
        ("depth_mode", C.c_int32),
    ]


class PgrOutputs(C.Structure):
    _fields_ = [("color", C.c_void_p), ("depth", C.c_void_p), ("radii", C.c_void_p), ("final_T", C.c_void_p),
                ("n_contrib", C.c_void_p), ("sem_color", C.c_void_p), ("sem_depth", C.c_void_p),
                ("sem_masks", C.c_void_p), ("record", C.c_void_p)]


class PgrSemantic(C.Structure):
    _fields_ = [("object_id", C.c_void_p), ("colors", C.c_void_p), ("n_env", C.c_int32), ("k_objects", C.c_int32),
                ("object_id_u8", C.c_void_p), ("mask_colors", C.c_void_p), ("mask_threshold", C.c_float)]


class PgrLayers(C.Structure):
    _fields_ = [("layer_id", C.c_void_p), ("n_layers", C.c_int32), ("mask_colors", C.c_void_p),
                ("mask_threshold", C.c_float)]


class PgrRecordLayout(C.Structure):
    _fields_ = [("off_rgb", C.c_int64), ("off_depth", C.c_int64), ("off_masks", C.c_int64), ("bytes", C.c_int64)]


PGR_DEPTH_EXPECTED = 0
PGR_DEPTH_NORMALIZED = 1


class PgrPosedObjects(C.Structure):
    _fields_ = [("object_id", C.c_void_p), ("poses", C.c_void_p), ("k_objects", C.c_int32)]


PGR_POSE_STRIDE = 20


class PgrObjectPose(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3), ("center", C.c_float * 3), ("q", C.c_float * 4),
                ("D1", C.c_float * 9), ("D2", C.c_float * 25), ("D3", C.c_float * 49)]


class PgrPoseJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R", C.c_void_p), ("t", C.c_void_p), ("n", C.c_int32),
                ("kind", C.c_int32), ("n_rest", C.c_int32), ("about_origin", C.c_int32), ("R_row_stride", C.c_int32),
                ("t_stride", C.c_int32)]


PGR_POSE_XYZ, PGR_POSE_ROT, PGR_POSE_SH = 0, 1, 2


class PgrGradOutputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("means2d", "means3d", "opacities", "colors", "shs", "cov3d", "scales",
                                          "rotations")]


class PgrWorkspaceView(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("splats", "rects", "gauss_sorted", "ranges", "num_instances")]


# every symbol include/pegasus_raster.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pgr_abi_version": (C.c_int32, []),
    "pgr_version": (C.c_char_p, []),
    "pgr_status_string": (C.c_char_p, [C.c_int32]),
    "pgr_last_hip_error": (C.c_char_p, []),
    "pgr_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "pgr_forward": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrCamera), C.POINTER(PgrOutputs), C.c_void_p,
                                C.c_size_t, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "pgr_batch_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    "pgr_forward_batch": (C.c_int32, [C.POINTER(PgrScene), C.c_int32, C.POINTER(PgrCamera), C.POINTER(PgrOutputs),
                                      C.c_void_p, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "pgr_host_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "pgr_forward_batch_async": (C.c_int32, [C.POINTER(PgrScene), C.c_int32, C.POINTER(PgrCamera),
                                            C.POINTER(PgrOutputs), C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "pgr_forward_frames_async": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrSemantic), C.c_int32,
                                             C.POINTER(PgrCamera), C.POINTER(PgrOutputs), C.c_void_p, C.c_size_t,
                                             C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgr_forward_posed_async": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrSemantic), C.POINTER(PgrPosedObjects),
                                            C.c_int32, C.POINTER(PgrCamera), C.POINTER(PgrOutputs), C.c_void_p,
                                            C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgr_forward_posed_early_status": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrSemantic), C.POINTER(PgrPosedObjects),
                                                   C.c_int32, C.POINTER(PgrCamera), C.POINTER(PgrOutputs), C.c_void_p,
                                                   C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pgr_batch_status": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "pgr_scene_cache_bytes": (C.c_size_t, [C.c_int32]),
    "pgr_scene_prepare": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrSemantic), C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "pgr_layers_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    "pgr_forward_layers_async": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrLayers), C.POINTER(PgrPosedObjects),
                                             C.c_int32, C.POINTER(PgrCamera), C.POINTER(PgrOutputs), C.c_void_p,
                                             C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgr_frame_record_layout": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(PgrRecordLayout)]),
    "pgr_pack_records": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_int64, C.c_void_p]),
    "pgr_forward_batch_profiled": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrSemantic), C.c_int32,
                                               C.POINTER(PgrCamera),
                                               C.POINTER(PgrOutputs), C.c_void_p, C.c_size_t, C.c_int64,
                                               C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_float)]),
    "pgr_workspace_view": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                       C.c_int32, C.c_int32, C.POINTER(PgrWorkspaceView)]),
    "pgr_backward": (C.c_int32, [C.POINTER(PgrScene), C.POINTER(PgrCamera), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.POINTER(PgrGradOutputs),
                                 C.c_void_p, C.c_void_p]),
    "pgr_compose_object": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.POINTER(PgrObjectPose), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p]),
    "pgr_clock_probe": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "pgr_pose_objects_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "pgr_pose_objects": (C.c_int32, [C.c_int32, C.POINTER(PgrPoseJob), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "pgr_block_visibility_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "pgr_block_visibility": (C.c_int32, [C.POINTER(PgrScene), C.c_int32, C.POINTER(PgrCamera), C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p]),
    "pgr_mark_visible": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgr_color_masks": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                    C.c_void_p, C.c_void_p]),
    "pgr_knn_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "pgr_knn_mean_dist2": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgr_quantize_frame": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "pgr_pack_frames": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class RasterizerLibraryError(RuntimeError):
    pass


def lib():
    """Loads the HIP library (once).  Raises RasterizerLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RasterizerLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m pegasus_amd.build` "
                "(there is no CPU fallback for the rasterizer)")
        # torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64, and the library's DT_NEEDED entries name the
        # same sonames -- loaded behind torch it binds to torch's runtime (one HIP runtime in the process, the one that owns
        # the tensors it is handed); loaded in front of it the system's runtime comes in as a SECOND one and every HIP call
        # of the library fails with "no ROCm-capable device is detected" (build() followed by smoke() in one process did)
        import torch  # noqa: F401
        try:
            handle = C.CDLL(str(LIB_PATH))
        except OSError as e:
            raise RasterizerLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        # the ABI version first, binding only that symbol: a stale or variant build (PGR_LIB) then fails with the
        # version message, not with an AttributeError on whichever newer entry point it lacks
        try:
            ver_fn = handle.pgr_abi_version
        except AttributeError as e:
            raise RasterizerLibraryError(f"{LIB_PATH} does not export pgr_abi_version: not a pegasus_raster library") from e
        ver_fn.restype, ver_fn.argtypes = C.c_int32, []
        if ver_fn() != PGR_ABI_VERSION:
            raise RasterizerLibraryError(f"{LIB_PATH}: ABI version {ver_fn()} but this package binds version "
                                         f"{PGR_ABI_VERSION}; rebuild with `python -m pegasus_amd.build`")
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RasterizerLibraryError(f"{LIB_PATH} does not export {name} (include/pegasus_raster.h)") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int, what: str = "pegasus_raster"):
    if status == PGR_OK:
        return
    L = lib()
    msg = f"{what}: {L.pgr_status_string(status).decode()} ({status})"
    if status == PGR_ERR_LAUNCH_FAILURE:
        msg += f" [{L.pgr_last_hip_error().decode()}]"
    if status == PGR_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise RuntimeError(msg)
