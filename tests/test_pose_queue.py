"""Deferred pose calls (pegasus_amd/pose_queue.py, pgr_pose_objects): the three pose methods PEGASUS calls per object and
frame (/root/reference/src/gs/pegasus_setup.py:195-208) recorded and applied in one device pass must give what applying
each call at once gives (rounds 1-5's path, itself pinned by tests/golden/gs_model.npz and pose_recursion.npz)."""
import copy

import numpy as np
import pytest
import torch


def _objects(dev, sizes=(700, 1, 2500), seed=5, n_rest=15):
    from pegasus_amd.gaussian_model import GaussianModel
    rng = np.random.default_rng(seed)
    out = []
    for n in sizes:
        m = GaussianModel(3, device=dev)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        m._xyz = t(rng.normal(0, 0.3, (n, 3)) + rng.normal(0, 1, 3))
        m._features_dc = t(rng.uniform(-1, 1, (n, 1, 3)))
        m._features_rest = t(rng.normal(0, 0.1, (n, n_rest, 3)))
        m._opacity = t(rng.normal(0, 1, (n, 1)))
        m._scaling = t(rng.normal(-4, 0.3, (n, 3)))
        m._rotation = t(rng.normal(0, 1, (n, 4)) * rng.uniform(0.5, 2.0, (n, 1)))
        out.append(m)
    return out


def _poses(dev, k, seed):
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(seed)
    res = []
    for i in range(k):
        R = torch.from_numpy(Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()).type(torch.float32).to(dev)
        t = torch.from_numpy(rng.normal(0, 0.2, 3)).type(torch.float32).to(dev)
        T = torch.eye(4, dtype=torch.float32, device=dev)
        T[:3, :3] = R
        T[:3, 3] = t
        res.append((R, t, T))
    return res


def _reference_calls(obj, R, T):                       # apply_transformation_on_gs, literally
    obj.apply_transformation_on_xyz(T=T)
    obj.apply_rotation_on_splats(R=R)
    obj.apply_rotation_on_sh(R=R)


def _same(a, b, what):
    np.testing.assert_allclose(a._xyz.cpu().numpy(), b._xyz.cpu().numpy(), atol=2e-6, err_msg=what)
    qa, qb = a._rotation.cpu().numpy().astype(np.float64), b._rotation.cpu().numpy().astype(np.float64)
    s = np.sign(np.sum(qa * qb, axis=1, keepdims=True))            # q and -q are one rotation
    np.testing.assert_allclose(qa * s, qb, atol=2e-6, err_msg=what)
    np.testing.assert_allclose(a._features_rest.cpu().numpy(), b._features_rest.cpu().numpy(), atol=2e-6, err_msg=what)


@pytest.mark.gpu
@pytest.mark.parametrize("n_rest", [15, 8, 3])
def test_recorded_pose_calls_equal_calls_applied_at_once(gpu_device, n_rest):
    from pegasus_amd import pose_queue as PQ
    objs, ref = _objects(gpu_device, n_rest=n_rest), _objects(gpu_device, n_rest=n_rest)
    before = PQ.set_enabled(False)
    try:
        for o, (R, t, T) in zip(ref, _poses(gpu_device, 3, 1)):
            _reference_calls(o, R, T)
        for o, (R, t, T) in zip(ref, _poses(gpu_device, 3, 2)):       # a second update on top
            _reference_calls(o, R, T)
    finally:
        PQ.set_enabled(before)
    assert PQ.enabled()
    first, second = _poses(gpu_device, 3, 1), _poses(gpu_device, 3, 2)
    for o, (R, t, T) in zip(objs, first):
        _reference_calls(o, R, T)
        R.zero_(); T.fill_(7.0)                                       # the caller's tensors may change: the queue owns copies
    assert all(len(o.__dict__["_pose_ops"]) == 4 for o in objs)       # nothing applied yet: rot_xyz, translate, splats, sh
    for o, (R, t, T) in zip(objs, second):                            # two updates pending per object: two rounds of jobs
        _reference_calls(o, R, T)
    _ = objs[1]._rotation                                             # ONE read applies every model's pending calls
    assert not any(o.__dict__.get("_pose_ops") for o in objs)
    for k, (a, b) in enumerate(zip(objs, ref)):
        _same(a, b, f"object {k}")


@pytest.mark.gpu
def test_every_way_of_looking_at_a_model_sees_the_recorded_calls(gpu_device):
    """Reads, writes, copy.copy, copy.deepcopy, merge_gaussians, mask_points and the fused one-call API all apply what is
    pending first; rotation about the origin, a lone translation and autograd-tracked models take their own paths."""
    from pegasus_amd import pose_queue as PQ
    (R, t, T), = _poses(gpu_device, 1, 3)

    def posed(fn):
        (a,), (b,) = _objects(gpu_device, (900,)), _objects(gpu_device, (900,))
        before = PQ.set_enabled(False)
        try:
            fn(b)
        finally:
            PQ.set_enabled(before)
        fn(a)
        return a, b
    a, b = posed(lambda o: _reference_calls(o, R, T))
    assert a.__dict__.get("_pose_ops")
    _same(copy.copy(a), b, "copy.copy")
    a, b = posed(lambda o: _reference_calls(o, R, T))
    _same(copy.deepcopy(a), b, "copy.deepcopy")
    a, b = posed(lambda o: _reference_calls(o, R, T))
    (env,) = _objects(gpu_device, (40,), seed=9)
    env.merge_gaussians(a)
    np.testing.assert_allclose(env._xyz[40:].cpu().numpy(), b._xyz.cpu().numpy(), atol=2e-6)
    a, b = posed(lambda o: o.apply_transformation(T))                 # the fused one-call API
    _same(a, b, "apply_transformation")
    a, b = posed(lambda o: (o.apply_rotation_on_xyz(R, origin=True), o.apply_translation_on_xyz(t)))
    _same(a, b, "about the origin")
    a, b = posed(lambda o: o.apply_translation_on_xyz(t))             # alone: applied at once, bit for bit
    assert not a.__dict__.get("_pose_ops") and torch.equal(a._xyz, b._xyz)
    a, b = posed(lambda o: (o.apply_rotation_on_splats(R), setattr(o, "_rotation", o._rotation * 2.0)))
    _same(a, b, "write after a recorded call")
    mask = torch.arange(900, device=gpu_device) % 3 == 0
    a, b = posed(lambda o: (_reference_calls(o, R, T), o.mask_points(mask)))
    _same(a, b, "mask_points")
    (g,) = _objects(gpu_device, (50,))
    g._xyz = g._xyz.clone().requires_grad_(True)
    g.apply_rotation_on_xyz(R)                                        # tracked by autograd: torch ops, at once
    assert not g.__dict__.get("_pose_ops") and g._xyz.grad_fn is not None


@pytest.mark.gpu
def test_pose_objects_abi_rejects_bad_jobs(gpu_device):
    import ctypes as C
    from pegasus_amd import _lib
    L = _lib.lib()
    x = torch.zeros((8, 3), device=gpu_device)
    ws = torch.empty(int(L.pgr_pose_objects_workspace_bytes(1)), dtype=torch.uint8, device=gpu_device)
    job = lambda **kw: (_lib.PgrPoseJob * 1)(_lib.PgrPoseJob(**kw))
    good = dict(src=x.data_ptr(), dst=x.data_ptr(), n=8, kind=_lib.PGR_POSE_XYZ)
    call = lambda j, w=ws: L.pgr_pose_objects(1, j, None, None, C.c_void_p(w.data_ptr()), int(w.numel()), None)
    assert call(job(**good)) == _lib.PGR_OK                                           # identity pose in place
    assert call(job(**dict(good, kind=7))) == _lib.PGR_ERR_INVALID_ARGUMENT
    assert call(job(**dict(good, dst=None))) == _lib.PGR_ERR_INVALID_ARGUMENT
    assert call(job(**dict(good, kind=_lib.PGR_POSE_SH, n_rest=15))) == _lib.PGR_ERR_INVALID_ARGUMENT    # no SH tables
    assert L.pgr_pose_objects(1, job(**good), None, None, C.c_void_p(ws.data_ptr()), 8, None) == _lib.PGR_ERR_WORKSPACE_TOO_SMALL
    assert L.pgr_pose_objects(0, None, None, None, None, 0, None) == _lib.PGR_OK
    torch.cuda.synchronize()
    assert float(x.abs().max()) == 0.0
