import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from pegasus_amd import scenes
from helpers import gpu_forward
n, spread = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(n)
cloud, views = scenes.scene_c1(seed=4, n=n)
cloud.xyz[:] = rng.normal(0, spread, size=(n, 3)).astype(np.float32)
cloud.scaling[:] = np.log(0.004).astype(np.float32)
cloud.opacity[:] = rng.normal(-4.0, 0.5, size=(n, 1)).astype(np.float32)
act = cloud.activated(); v = views[0]
o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs(), num_threads=8, cull_mode=1)
g = gpu_forward(act, v, sh_degree=3, device="cuda:0")
r = o["ranges"]
for t in range(r.shape[0]):
    a, b = r[t]
    if b - a == 0: continue
    G, O = g["gauss_sorted"][a:b], o["gauss_sorted"][a:b]
    if not np.array_equal(G, O):
        d = g["depth"]
        print("tile", t, "len", b - a, "perm ok", np.array_equal(np.sort(G), np.sort(O)),
              "first bad", int(np.nonzero(G != O)[0][0]), "n bad", int((G != O).sum()))
        dg = d[G].view(np.uint32).astype(np.int64)
        bad = np.nonzero(np.diff(dg) < 0)[0]
        print("  depth inversions at", bad[:20], "count", bad.size)
        if bad.size: print("  around first inversion", dg[max(0,bad[0]-2):bad[0]+4] - dg.min())
        uniq, cnt = np.unique(G, return_counts=True)
        print("  duplicates", int((cnt > 1).sum()), "missing", int(np.setdiff1d(O, G).size))
L = (r[:, 1] - r[:, 0]).astype(np.int64)
print("n", n, "max list", L.max(), "lists>8192", int((L > 8192).sum()), "lists>16384", int((L > 16384).sum()),
      "all equal", np.array_equal(g["gauss_sorted"], o["gauss_sorted"]))
