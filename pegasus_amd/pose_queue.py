"""Deferred object poses: the pose calls of a frame as ONE device pass (pgr_pose_objects).

PEGASUS moves every object between two frames of a dynamic sequence with three calls per object
(/root/reference/src/gs/pegasus_setup.py:178-208 -> gaussian_model.py:482-546): apply_transformation_on_xyz(T),
apply_rotation_on_splats(R), apply_rotation_on_sh(R), with T and R built on the device.  Applied one by one each call costs
launches and -- for the quaternion and the SH band matrices, which rounds 1-5 derived on the host -- a device round trip:
2.9 ms per frame for 8 objects (profiles/r05_bench_facade.json).  Nothing reads an object between these calls, so
GaussianModel RECORDS them (a private copy of the small pose tensor each) and applies what is pending when one of the three
attributes they change (_xyz, _rotation, _features_rest) is next read -- of ANY model: one flush serves every object with
pending calls, in one pgr_pose_objects call per round of independent jobs (three kernel launches per 16 jobs; the cloud mean,
quat(R) and the SH band matrices are computed on the device from the recorded tensors).

Semantics are those of applying every call at once: a call sees the arrays as the previous calls left them (an attribute
read or written in between flushes first), results replace the attribute with a NEW tensor exactly as the reference's
methods do, and autograd-tracked or host-resident models never defer."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib

ROT_XYZ, TRANSLATE, ROT_SPLATS, ROT_SH = "rot_xyz", "translate", "rot_splats", "rot_sh"
_ARRAY = {ROT_XYZ: "_xyz", TRANSLATE: "_xyz", ROT_SPLATS: "_rotation", ROT_SH: "_features_rest"}

_pending = {}          # id(model) -> weakref(model): the models with recorded calls
_tables = {}           # device -> (sh_dirs, sh_pinv) fp64 device tensors
_workspace = {}        # device -> uint8 tensor
_last_clone = [None]   # (key, source tensor, private copy): PEGASUS hands the same R to two calls in a row


def enabled() -> bool:
    return _enabled[0]


_enabled = [True]


def set_enabled(on: bool) -> bool:
    """Switch deferral off (every pose call applies at once, rounds 1-5) or on; returns the previous setting."""
    before = _enabled[0]
    if not on:
        flush_all()
    _enabled[0] = bool(on)
    return before


def private_copy(M: torch.Tensor) -> torch.Tensor:
    """float32 device copy of a pose tensor that the queue owns: the caller may change or free its tensor before the
    recorded call is applied.  The copy of an unchanged tensor handed over twice in a row is shared."""
    try:
        key = (id(M), M._version, M.data_ptr(), tuple(M.shape))
    except RuntimeError:
        key = None
    hit = _last_clone[0]
    if key is not None and hit is not None and hit[0] == key:
        return hit[2]
    c = M.detach().to(torch.float32).clone(memory_format=torch.contiguous_format)
    _last_clone[0] = (key, M, c)           # (keeps M alive: its id cannot be recycled under the key)
    return c


def record(model, kind: str, R=None, t=None, origin: bool = False, R_stride: int = 3, t_stride: int = 1):
    ops = model.__dict__.setdefault("_pose_ops", [])
    ops.append((kind, R, t, bool(origin), int(R_stride), int(t_stride)))
    _pending[id(model)] = weakref.ref(model)


def _sh_tables(device):
    hit = _tables.get(device)
    if hit is None:
        from . import sh_rotation as SR
        pinv = np.concatenate([SR._PINV[1], SR._PINV[2], SR._PINV[3]], axis=0)        # [15, 61]
        assert pinv.shape == (15, 61) and SR._DIRS.shape == (61, 3)
        hit = _tables[device] = (torch.from_numpy(np.ascontiguousarray(SR._DIRS, dtype=np.float64)).to(device),
                                 torch.from_numpy(np.ascontiguousarray(pinv, dtype=np.float64)).to(device))
    return hit


def flush_all():
    """Applies every recorded pose call of every model (rounds of mutually independent jobs)."""
    if not _pending:
        return
    models = []
    for ref in list(_pending.values()):
        m = ref()
        if m is not None and m.__dict__.get("_pose_ops"):
            models.append(m)
    _pending.clear()
    L = _lib.lib()
    while True:
        jobs, outs, keep = [], [], []
        for m in models:
            ops = m.__dict__["_pose_ops"]
            used = set()
            while ops:
                kind, R, t, origin, rs, ts = ops[0]
                arr = _ARRAY[kind]
                if arr in used:
                    break
                used.add(arr)
                ops.pop(0)
                src = m.__dict__[arr].contiguous()
                if kind == ROT_XYZ and ops and ops[0][0] == TRANSLATE:
                    # x' = R (x - c) + c, then + t: one job (what apply_transformation_on_xyz records)
                    _, _, t, _, _, ts = ops.pop(0)
                dst = torch.empty_like(src)
                code = {ROT_XYZ: _lib.PGR_POSE_XYZ, TRANSLATE: _lib.PGR_POSE_XYZ, ROT_SPLATS: _lib.PGR_POSE_ROT,
                        ROT_SH: _lib.PGR_POSE_SH}[kind]
                n_rest = int(src.shape[1]) if kind == ROT_SH else 0
                if kind == ROT_SH and n_rest not in (3, 8, 15):
                    if n_rest == 0:
                        continue
                    raise ValueError("_features_rest must hold 3, 8 or 15 coefficients per Gaussian")
                jobs.append(_lib.PgrPoseJob(src=src.data_ptr(), dst=dst.data_ptr(), R=R.data_ptr() if R is not None else None,
                                            t=t.data_ptr() if t is not None else None, n=int(src.shape[0]), kind=code,
                                            n_rest=n_rest, about_origin=1 if origin else 0, R_row_stride=rs, t_stride=ts))
                outs.append((m, arr, dst))
                keep.append((src, R, t))
        if not jobs:
            break
        for dev in sorted({o[2].device for o in outs}, key=str):             # one call per device that has jobs this round
            sel = [i for i, o in enumerate(outs) if o[2].device == dev]
            need = int(L.pgr_pose_objects_workspace_bytes(len(sel)))
            ws = _workspace.get(dev)
            if ws is None or ws.numel() < need:
                ws = _workspace[dev] = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=dev)
            dirs, pinv = _sh_tables(dev)
            arr_t = (_lib.PgrPoseJob * len(sel))(*[jobs[i] for i in sel])
            with torch.cuda.device(dev):
                _lib.check(L.pgr_pose_objects(len(sel), arr_t, C.c_void_p(dirs.data_ptr()), C.c_void_p(pinv.data_ptr()),
                                              C.c_void_p(ws.data_ptr()), int(ws.numel()),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "pgr_pose_objects")
        for m, arr, dst in outs:
            m.__dict__[arr] = dst
        # (`keep` dies here: the caching allocator hands the sources' blocks out again in stream order, behind the launches)
    for m in models:
        m.__dict__.pop("_pose_ops", None)
