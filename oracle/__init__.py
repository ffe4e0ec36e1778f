"""ctypes binding of the CPU oracle (oracle/pgr_oracle.c).

TEST INFRASTRUCTURE ONLY -- the product package ``pegasus_amd`` must never import this module.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

PARITY UNPINNED: the reference's rasterizer source is absent (/root/reference/.gitmodules:1-3);
see oracle/pgr_oracle.h for what the restatement is anchored on.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libpgr_oracle.so"
_CFLAGS = ["-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp", "-fPIC"]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (same flags as oracle/Makefile)."""
    srcs = [_HERE / "pgr_oracle.c", _HERE / "pgr_oracle_backward.c"]
    newest = max(p.stat().st_mtime for p in srcs + [_HERE / "pgr_oracle.h"])
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < newest:
        tmp = _LIB_PATH.with_suffix(f".tmp{os.getpid()}.so")
        subprocess.run(["gcc", *_CFLAGS, "-shared", "-o", str(tmp), *map(str, srcs), "-lm"], check=True)
        os.replace(tmp, _LIB_PATH)
    return _LIB_PATH


class _In(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("means3d", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3d_precomp", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p),
        ("sh_degree", C.c_int32), ("sh_stride", C.c_int32), ("scale_modifier", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
        ("campos", C.c_float * 3), ("bg", C.c_float * 3), ("cull_mode", C.c_int32),
        ("object_id", C.c_void_p), ("poses", C.c_void_p), ("k_objects", C.c_int32),
        ("tie_index", C.c_void_p), ("depth_mode", C.c_int32),
    ]


class _Out(C.Structure):
    _fields_ = [
        ("radii", C.c_void_p), ("tiles_touched", C.c_void_p), ("xy", C.c_void_p), ("depth", C.c_void_p),
        ("conic_opacity", C.c_void_p), ("rgb", C.c_void_p), ("cov3d", C.c_void_p),
        ("num_instances", C.c_int64), ("keys_sorted", C.c_void_p), ("gauss_sorted", C.c_void_p),
        ("cap_instances", C.c_int64), ("ranges", C.c_void_p),
        ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("final_T", C.c_void_p),
        ("n_contrib", C.c_void_p), ("ambig", C.c_void_p),
    ]


class _Grads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("means2d", "means3d", "opacities", "colors", "shs", "cov3d", "scales",
                                          "rotations")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.pgr_oracle_forward.argtypes = [C.POINTER(_In), C.POINTER(_Out), C.c_int]
        _lib.pgr_oracle_forward.restype = C.c_int
        _lib.pgr_oracle_preprocess.argtypes = [C.POINTER(_In), C.POINTER(_Out), C.c_int]
        _lib.pgr_oracle_preprocess.restype = C.c_int
        _lib.pgr_oracle_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.pgr_oracle_mark_visible.restype = C.c_int
        _lib.pgr_oracle_color_masks.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                C.c_float, C.c_void_p]
        _lib.pgr_oracle_color_masks.restype = C.c_int
        _lib.pgr_oracle_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _lib.pgr_oracle_quantize.restype = C.c_int
        _lib.pgr_oracle_tile_may_contribute.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                         C.c_int32]
        _lib.pgr_oracle_tile_may_contribute.restype = C.c_int
        _lib.pgr_oracle_backward.argtypes = [C.POINTER(_In), C.c_void_p, C.c_void_p, C.POINTER(_Grads), C.c_int]
        _lib.pgr_oracle_backward.restype = C.c_int
        _lib.pgr_oracle_version.restype = C.c_char_p
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _make_in(means3d, opacities, *, scales=None, rotations=None, cov3d_precomp=None, shs=None,
             colors_precomp=None, sh_degree=0, scale_modifier=1.0, width, height, tanfovx, tanfovy,
             viewmatrix, projmatrix, campos, bg, cull_mode=0, object_id=None, poses=None, tie_index=None,
             depth_mode=0):
    means3d = _f32(means3d).reshape(-1, 3)
    n = means3d.shape[0]
    keep = dict(means3d=means3d, opacities=_f32(opacities).reshape(-1) if n or opacities is not None else None,
                scales=_f32(scales), rotations=_f32(rotations), cov3d=_f32(cov3d_precomp), shs=_f32(shs),
                colors=_f32(colors_precomp))
    i = _In()
    i.n = n
    i.means3d = _ptr(keep["means3d"]); i.opacities = _ptr(keep["opacities"])
    i.scales = _ptr(keep["scales"]); i.rotations = _ptr(keep["rotations"])
    i.cov3d_precomp = _ptr(keep["cov3d"]); i.shs = _ptr(keep["shs"]); i.colors_precomp = _ptr(keep["colors"])
    i.sh_degree = int(sh_degree)
    i.sh_stride = int(keep["shs"].shape[1]) if keep["shs"] is not None and keep["shs"].ndim == 3 else 0
    i.scale_modifier = float(scale_modifier)
    i.width, i.height = int(width), int(height)
    i.tanfovx, i.tanfovy = float(tanfovx), float(tanfovy)
    i.viewmatrix = (C.c_float * 16)(*_f32(viewmatrix).reshape(16))
    i.projmatrix = (C.c_float * 16)(*_f32(projmatrix).reshape(16))
    i.campos = (C.c_float * 3)(*_f32(campos).reshape(3))
    i.bg = (C.c_float * 3)(*_f32(bg).reshape(3))
    i.cull_mode = int(cull_mode)
    i.depth_mode = int(depth_mode)
    if object_id is not None:      # posed objects: poses [K, 20] (pegasus_amd.compose.pose_table)
        keep["object_id"] = np.ascontiguousarray(np.asarray(object_id, dtype=np.int32))
        keep["poses"] = _f32(poses).reshape(-1, 20)
        i.object_id, i.poses, i.k_objects = _ptr(keep["object_id"]), _ptr(keep["poses"]), keep["poses"].shape[0]
    if tie_index is not None:
        keep["tie_index"] = np.ascontiguousarray(np.asarray(tie_index, dtype=np.int32))
        i.tie_index = _ptr(keep["tie_index"])
    return i, keep, n


def forward(means3d, opacities, *, num_threads=1, want_binning=True, stage="all", **kw):
    """Run the oracle.  Returns a dict of numpy arrays (all intermediates + images).

    ``stage='preprocess'`` stops after the per-Gaussian stage.
    """
    i, keep, n = _make_in(means3d, opacities, **kw)
    W, H = i.width, i.height
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    r = dict(
        radii=np.zeros(n, np.int32), tiles_touched=np.zeros(n, np.int32), xy=np.zeros((n, 2), np.float32),
        depth=np.zeros(n, np.float32), conic_opacity=np.zeros((n, 4), np.float32),
        rgb=np.zeros((n, 3), np.float32), cov3d=np.zeros((n, 6), np.float32),
    )
    o = _Out()
    for k in ("radii", "tiles_touched", "xy", "depth", "conic_opacity", "rgb", "cov3d"):
        setattr(o, k, _ptr(r[k]))
    if stage == "preprocess":
        rc = lib().pgr_oracle_preprocess(C.byref(i), C.byref(o), int(num_threads))
        if rc:
            raise ValueError(f"pgr_oracle_preprocess failed: {rc}")
        return r
    r.update(
        color=np.zeros((3, H, W), np.float32), out_depth=np.zeros((1, H, W), np.float32),
        final_T=np.zeros((H, W), np.float32), n_contrib=np.zeros((H, W), np.uint32),
        ambig=np.zeros((H, W), np.uint8), ranges=np.zeros((tiles, 2), np.uint32),
    )
    o.out_color = _ptr(r["color"]); o.out_depth = _ptr(r["out_depth"]); o.final_T = _ptr(r["final_T"])
    o.n_contrib = _ptr(r["n_contrib"]); o.ambig = _ptr(r["ambig"]); o.ranges = _ptr(r["ranges"])
    if want_binning:
        # size the instance arrays from a preprocess-only pass
        o2 = _Out()
        tt = np.zeros(n, np.int32)
        o2.tiles_touched = _ptr(tt)
        if n:
            lib().pgr_oracle_preprocess(C.byref(i), C.byref(o2), int(num_threads))
        cap = int(tt.sum(dtype=np.int64))
        r["keys_sorted"] = np.zeros(cap, np.uint64)
        r["gauss_sorted"] = np.zeros(cap, np.uint32)
        o.keys_sorted = _ptr(r["keys_sorted"]); o.gauss_sorted = _ptr(r["gauss_sorted"])
        o.cap_instances = cap
    rc = lib().pgr_oracle_forward(C.byref(i), C.byref(o), int(num_threads))
    if rc:
        raise ValueError(f"pgr_oracle_forward failed: {rc}")
    r["num_instances"] = int(o.num_instances)
    if want_binning:   # cull_mode 1 emits fewer instances than the rectangle count the arrays were sized for
        r["keys_sorted"] = r["keys_sorted"][:r["num_instances"]]
        r["gauss_sorted"] = r["gauss_sorted"][:r["num_instances"]]
    del keep
    return r


def backward(means3d, opacities, grad_color, grad_depth=None, *, num_threads=1, **kw):
    """Gradients of sum(grad_color * color) + sum(grad_depth * depth) wrt the rasterizer inputs (dict of numpy)."""
    i, keep, n = _make_in(means3d, opacities, **kw)
    H, W = i.height, i.width
    gc = _f32(grad_color).reshape(3, H, W)
    gd = None if grad_depth is None else _f32(grad_depth).reshape(H, W)
    stride = i.sh_stride
    r = dict(means2d=np.zeros((n, 3), np.float32), means3d=np.zeros((n, 3), np.float32),
             opacities=np.zeros(n, np.float32), colors=np.zeros((n, 3), np.float32),
             cov3d=np.zeros((n, 6), np.float32))
    if keep["shs"] is not None:
        r["shs"] = np.zeros((n, stride, 3), np.float32)
    if keep["scales"] is not None:
        r["scales"] = np.zeros((n, 3), np.float32)
        r["rotations"] = np.zeros((n, 4), np.float32)
    g = _Grads()
    for k, v in r.items():
        setattr(g, k, _ptr(v))
    rc = lib().pgr_oracle_backward(C.byref(i), _ptr(gc), _ptr(gd), C.byref(g), int(num_threads))
    if rc:
        raise ValueError(f"pgr_oracle_backward failed: {rc}")
    del keep
    return r


def mark_visible(means3d, viewmatrix):
    m = _f32(means3d).reshape(-1, 3)
    v = _f32(viewmatrix).reshape(16)
    out = np.zeros(m.shape[0], np.uint8)
    rc = lib().pgr_oracle_mark_visible(m.shape[0], _ptr(m), _ptr(v), _ptr(out))
    if rc:
        raise ValueError("pgr_oracle_mark_visible failed")
    return out.astype(bool)


def color_masks(img_chw, colors, thr=0.1):
    img = _f32(img_chw)
    colors = _f32(colors).reshape(-1, 3)
    _, H, W = img.shape
    out = np.zeros((colors.shape[0], H, W), np.uint8)
    rc = lib().pgr_oracle_color_masks(_ptr(img), W, H, _ptr(colors), colors.shape[0], float(thr), _ptr(out))
    if rc:
        raise ValueError("pgr_oracle_color_masks failed")
    return out


def quantize(img_chw, depth_hw):
    img = _f32(img_chw)
    d = _f32(depth_hw).reshape(img.shape[1], img.shape[2])
    _, H, W = img.shape
    rgb = np.zeros((H, W, 3), np.uint8)
    mm = np.zeros((H, W), np.uint16)
    rc = lib().pgr_oracle_quantize(_ptr(img), _ptr(d), W, H, _ptr(rgb), _ptr(mm))
    if rc:
        raise ValueError("pgr_oracle_quantize failed")
    return rgb, mm


def tile_may_contribute(xy, conic_opacity, tx, ty, width, height) -> bool:
    a = _f32(xy).reshape(2)
    b = _f32(conic_opacity).reshape(4)
    return bool(lib().pgr_oracle_tile_may_contribute(_ptr(a), _ptr(b), int(tx), int(ty), int(width), int(height)))


def version() -> str:
    return lib().pgr_oracle_version().decode()
