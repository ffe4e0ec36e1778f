"""Training-path timing: loss.backward() through the drop-in GaussianRasterizer (forward + HIP backward), and a
gradient check against the oracle at full image size."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from pegasus_amd import diff_gaussian_rasterization as dgr, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
if which == "c2":
    cloud, views = scenes.scene_c2(n=150_000, n_views=4)
else:
    cloud, views = scenes.scene_c3(scale=float(sys.argv[2]) if len(sys.argv) > 2 else 0.25, n_views=4)
a = cloud.activated()
tt = lambda arr, rg=True: torch.from_numpy(np.ascontiguousarray(arr)).to(dev).requires_grad_(rg)
means, op, sc, rot, shs = tt(a["means3d"]), tt(a["opacities"].reshape(-1, 1)), tt(a["scales"]), tt(a["rotations"]), tt(a["shs"])


def run(v, check=False):
    s = dgr.GaussianRasterizationSettings(v.height, v.width, v.tanfovx, v.tanfovy, torch.zeros(3, device=dev), 1.0,
                                          tt(v.world_view_transform, False), tt(v.full_proj_transform, False), 3,
                                          tt(v.camera_center, False), False, False)
    means2d = torch.zeros_like(means, requires_grad=True)
    color, radii, depth = dgr.GaussianRasterizer(s)(means, means2d, op, shs=shs, scales=sc, rotations=rot)
    loss = (color * wC).sum() + (depth[0] * wD).sum()
    loss.backward()
    return means2d


v0 = views[0]
rng = np.random.default_rng(0)
gC = rng.uniform(-1, 1, size=(3, v0.height, v0.width)).astype(np.float32)
gD = rng.uniform(-1, 1, size=(v0.height, v0.width)).astype(np.float32)
wC, wD = tt(gC, False), tt(gD, False)
for t in (means, op, sc, rot, shs):
    t.grad = None
m2d = run(v0)
torch.cuda.synchronize()
got = dict(means3d=means.grad.clone(), opacities=op.grad.reshape(-1).clone(), scales=sc.grad.clone(),
           rotations=rot.grad.clone(), shs=shs.grad.clone(), means2d=m2d.grad.clone())
for i in range(3):
    run(views[i % len(views)])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for i in range(K):
    run(views[i % len(views)])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"{which}: N={cloud.n} {v0.width}x{v0.height}: forward+backward {dt*1e3:.2f} ms/iteration ({1/dt:.1f} it/s)")
if len(sys.argv) > 3 or which == "c2":
    import oracle
    t0 = time.perf_counter()
    g = oracle.backward(**a, sh_degree=3, grad_color=gC, grad_depth=gD, **v0.raster_kwargs(), num_threads=64)
    print(f"oracle backward {time.perf_counter()-t0:.1f} s")
    for k, tg in got.items():
        ref = g[k]
        err = np.abs(tg.cpu().numpy() - ref).max()
        scale = max(1e-6, np.abs(ref).max())
        print(f"  grad {k:10s} max|err| / max|ref| = {err/scale:.2e}   (max|ref| {scale:.3e})")
