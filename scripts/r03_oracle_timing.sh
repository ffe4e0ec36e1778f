cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import oracle
from pegasus_amd import scenes
oracle.build()
cloud, views = scenes.scene_c3(n_views=2)
act = cloud.activated()
kw = views[0].raster_kwargs()
for thr in (32, 64, 128, 256):
    ts = []
    for _ in range(3):
        t = time.perf_counter(); oracle.forward(**act, sh_degree=3, **kw, num_threads=thr, want_binning=False); ts.append(round(time.perf_counter() - t, 3))
    print("threads", thr, ts, flush=True)
PY
