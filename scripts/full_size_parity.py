"""More full-size views against the oracle than the test suite affords: C3 (2 M Gaussians) and C5 (5 M) at 800x800 through the
batch path (Morton-ordered resident scene, caller's tie order), per view: radii, per-tile lists and n_contrib bit-exact, images
within 1e-4 outside the oracle's ambiguity mask.   python scripts/full_size_parity.py [n_c3_views] [n_c5_views]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from helpers import fetch_workspace
from pegasus_amd import rasterizer as R, scenes
from pegasus_amd.frames import FrameRenderer

oracle.build()
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n5 = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bad = 0
# cameras spread over each config's own set (512 / 200 Fibonacci hemisphere views, ordered by elevation: the first ones alone
# would be the grazing views), the lowest one included
for name, maker, total, nv in (("C3", lambda n: scenes.scene_c3(n_views=n), 512, n3), ("C5", lambda n: scenes.scene_c5(n_views=n), 200, n5)):
    if nv <= 0:
        continue
    cloud, all_views = maker(total)
    views = [all_views[(k * (total - 1)) // max(nv - 1, 1)] for k in range(nv)]
    act = cloud.activated()
    fr = FrameRenderer(act["means3d"], act["opacities"], act["scales"], act["rotations"], act["shs"], cloud.object_id, device="cuda:0")
    act_r = {k: np.ascontiguousarray(a[fr.order]) for k, a in act.items()}
    specs = [fr.view_spec(v) for v in views[:nv]]
    res = R.forward_views(fr.means3d, fr.opacities, specs, shs=fr.shs, scales=fr.scales, rotations=fr.rotations, sh_degree=3,
                          want_radii=True, want_aux=True, tie_index=fr.tie_index)
    torch.cuda.synchronize()
    for i, v in enumerate(views[:nv]):
        t0 = time.perf_counter()
        o = oracle.forward(**act_r, sh_degree=3, **v.raster_kwargs(), num_threads=32, cull_mode=1, tie_index=fr.order)
        w = fetch_workspace(i, cloud.n, v.width, v.height)
        amb = o["ambig"].astype(bool)
        fails = []
        if not np.array_equal(res[i]["radii"].cpu().numpy(), o["radii"]): fails.append("radii")
        if not np.array_equal(w["gauss_sorted"], o["gauss_sorted"]): fails.append("lists")
        if not np.array_equal(res[i]["n_contrib"].cpu().numpy()[~amb], o["n_contrib"][~amb]): fails.append("n_contrib")
        dc = np.abs(res[i]["color"].cpu().numpy() - o["color"])[:, ~amb].max()
        dd = np.abs(res[i]["depth"].cpu().numpy() - o["out_depth"])[:, ~amb].max()
        if dc > 1e-4 or dd > 1e-4: fails.append(f"image {dc:.2e} {dd:.2e}")
        bad += bool(fails)
        print(f"{name} view {i}: instances {o['num_instances']}, longest list {int((o['ranges'][:, 1] - o['ranges'][:, 0]).max())}, "
              f"ambiguous pixels {amb.mean():.1e}, max |dcolor| {dc:.1e}  {'MISMATCH ' + str(fails) if fails else 'ok'}  "
              f"(oracle {time.perf_counter() - t0:.1f} s)", flush=True)
    del fr, res
    torch.cuda.empty_cache()
print(f"{n3} C3 + {n5} C5 full-size views: {bad} with a mismatch")
