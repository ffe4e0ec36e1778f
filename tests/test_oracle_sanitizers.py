"""The CPU oracle (the checker every parity claim rests on) under AddressSanitizer + UndefinedBehaviorSanitizer: a sanitised build of
oracle/*.c runs a spread of forward calls -- both list modes, a ragged image size, posed objects, an empty scene -- and the backward
oracle in a subprocess with libasan / libubsan preloaded; any report fails the test.  (GPU sanitizers are not available on the pool;
the HIP path is covered by the fuzz and hostile-input suites instead.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

_SCRIPT = r'''
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, %(root)r)
import oracle
oracle._LIB_PATH = Path(%(lib)r)
oracle.build = lambda force=False: oracle._LIB_PATH
from pegasus_amd import scenes
for mk in (lambda: scenes.scene_c1(), lambda: scenes.scene_c3(scale=0.01, n_views=3, width=200, height=136),
           lambda: scenes.scene_c3(scale=0.005, n_views=2, width=37, height=19)):
    cloud, views = mk()
    act = cloud.activated()
    for v in views[:2]:
        for mode in (0, 1):
            o = oracle.forward(**act, sh_degree=3, **v.raster_kwargs((0.1, 0.2, 0.3)), num_threads=4, cull_mode=mode)
        assert o["num_instances"] > 0 and np.isfinite(o["color"]).all()
cloud, views, rest = scenes.merged_scene(7, 3000, 3, 500, 2, 96, 64)
act = cloud.activated()
poses = np.zeros((3, 20), np.float32); poses[:, 0] = poses[:, 4] = poses[:, 8] = 1.0; poses[:, 15] = 1.0; poses[:, 9] = 0.01
o = oracle.forward(**act, sh_degree=3, **views[0].raster_kwargs(), num_threads=2, object_id=cloud.object_id, poses=poses,
                   tie_index=np.random.default_rng(0).permutation(cloud.n).astype(np.int32))
z = lambda *s: np.zeros(s, np.float32)
e = oracle.forward(z(0, 3), z(0), scales=z(0, 3), rotations=z(0, 4), shs=z(0, 16, 3), sh_degree=3, **views[0].raster_kwargs(), num_threads=2)
assert e["num_instances"] == 0
m = oracle.color_masks(o["color"], np.array([[0.1, 0.2, 0.3], [0.9, 0.9, 0.9]], np.float32), 0.1)
# the backward oracle (pgr_oracle_backward.c) on the same small merged scene: every gradient finite
rng = np.random.default_rng(3)
H, W = views[0].height, views[0].width
g = oracle.backward(**act, grad_color=rng.normal(size=(3, H, W)).astype(np.float32), grad_depth=rng.normal(size=(H, W)).astype(np.float32),
                    sh_degree=3, **views[0].raster_kwargs(), num_threads=2)
assert all(np.isfinite(v).all() for v in g.values()) and float(np.abs(g["means3d"]).max()) > 0
print("SANITIZED RUN OK", int(o["num_instances"]), int(m.sum()))
'''


@pytest.mark.timeout(600)
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip("gcc's libasan / libubsan are not installed")
    lib = tmp_path / "libpgr_oracle_san.so"
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp", "-fPIC",
                    "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared", "-o", str(lib),
                    str(ROOT / "oracle" / "pgr_oracle.c"), str(ROOT / "oracle" / "pgr_oracle_backward.c"), "-lm"], check=True)
    env = dict(os.environ, LD_PRELOAD=f"{asan} {ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": str(ROOT), "lib": str(lib)}], env=env, capture_output=True,
                         text=True, timeout=560)
    report = out.stdout + out.stderr
    assert "SANITIZED RUN OK" in out.stdout, report[-3000:]
    assert "AddressSanitizer" not in report and "runtime error" not in report, report[-3000:]
