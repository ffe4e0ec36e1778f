"""Longer run of tests/test_fuzz_parity.py's random cases:  python scripts/fuzz_parity.py [first] [count]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle
import test_fuzz_parity as T
from helpers import gpu_forward

oracle.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = 0
for seed in range(first, first + count):
    act, view, deg, mod, bg = T._case(seed)
    o = oracle.forward(**act, sh_degree=deg, **view.raster_kwargs(bg), num_threads=8, scale_modifier=mod, cull_mode=1)
    g = gpu_forward(act, view, sh_degree=deg, bg=bg, device="cuda:0", scale_modifier=mod)
    amb = o["ambig"].astype(bool)
    ok = ~amb & np.isfinite(o["color"]).all(axis=0) & np.isfinite(o["out_depth"][0])
    fails = []
    if not np.array_equal(g["radii"], o["radii"]): fails.append("radii")
    if g["num_instances"] != o["num_instances"]: fails.append("instances")
    elif not np.array_equal(g["gauss_sorted"], o["gauss_sorted"]): fails.append("lists")
    if not np.array_equal(g["n_contrib"][ok], o["n_contrib"][ok]): fails.append("n_contrib")
    err = (np.abs(g["color"] - o["color"]) / np.maximum(1.0, np.abs(o["color"])))[:, ok].max(initial=0)
    if err > 1e-4: fails.append(f"colour {err:.2e}")
    o0 = oracle.forward(**act, sh_degree=deg, **view.raster_kwargs(bg), num_threads=8, scale_modifier=mod, cull_mode=0)   # reference-style lists
    ok0 = ~o0["ambig"].astype(bool) & np.isfinite(o0["color"]).all(axis=0) & np.isfinite(o0["out_depth"][0])
    if not np.array_equal(g["radii"], o0["radii"]): fails.append("radii vs reference-style")
    err0 = (np.abs(g["color"] - o0["color"]) / np.maximum(1.0, np.abs(o0["color"])))[:, ok0].max(initial=0)
    if err0 > 1e-4: fails.append(f"colour vs reference-style lists {err0:.2e}")
    from test_gpu_parity import _last_blended
    if not np.array_equal(_last_blended(g["gauss_sorted"], g["ranges"], g["n_contrib"], view.width, view.height)[ok0],
                          _last_blended(o0["gauss_sorted"], o0["ranges"], o0["n_contrib"], view.width, view.height)[ok0]):
        fails.append("last blended Gaussian vs reference-style lists")
    if fails:
        bad += 1
        print(f"seed {seed}: n {act['means3d'].shape[0]} {view.width}x{view.height} deg {deg} mod {mod}: {fails}")
print(f"{count} random cases from seed {first}: {bad} with a mismatch")
