"""bench.py's N > 1 path without a GPU: `python bench.py --gpus 2` must start its own two ranks (no outer wrapper), report
the world size it actually initialised, run the per-batch gather inside the timed region and verify what arrived.  The
frame source is bench.py's --stub-renderer (the rasterizer has no CPU path by design); everything else -- launcher,
argument handling, sharding, asynchronous gather, gather check, timing protocol, JSON line -- is the code the GPU run uses."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=str(ROOT), env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_gpus_2_launches_two_ranks_by_itself_and_gathers():
    p, line = _run([sys.executable, "bench.py", "--gpus", "2", "--stub-renderer", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1          # ONE line, from rank 0
    assert [l for l in p.stdout.splitlines() if l.strip()] == [l for l in p.stdout.splitlines() if l.startswith("{")], \
        "stdout carries the JSON line and nothing else (gloo / RCCL banners go to stderr)"
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    d = line["config"]["distributed"]
    assert d["world_size"] == 2 and d["launcher"].startswith("self")
    assert [x["rank"] for x in d["devices"]] == [0, 1]
    g = line["config"]["gather"]
    from pegasus_amd import masks as M
    assert g["check"] == "ok" and g["bytes_per_rank_and_batch"] == 4 * M.record_layout(4, 5, 8)["bytes"]
    assert g["views_per_s_with_gather"] == line["value"] and g["views_per_s_render_only"] > 0
    assert line["value"] > 0 and abs(line["ms_per_step"] * 3 * line["value"] / 1e3 - 3 * 4 * 2) < 0.05
    assert line.get("stub") is True and "INVALID" in line["metric"]
    # measurement protocol: R repeats of the K steps, the median one is the value; every rank's own time is in the line
    assert line["repeats"] == 5 and len(line["ms_per_step_all"]) == 5
    assert line["value_min"] <= line["value"] <= line["value_max"]
    assert sorted(line["ms_per_step_all"])[2] == line["ms_per_step"]
    pr = line["per_rank_s"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["rank0"] <= pr["max"] + 1e-9 and pr["max"] * 1e3 <= line["ms_per_step"] * 3 + 1e-3
    assert g["rank0_gather_host_ms_per_step"] >= 0


import pytest as _pytest


@_pytest.mark.parametrize("world", [4, 8])
def test_stub_gather_at_world_4_and_8(world):
    """The N-rank protocol at the node's real sizes (gloo, stub frame source): one gather per batch into the rank-major
    buffer, content checked through the (rank, i) -> global id rule."""
    p, line = _run([sys.executable, "bench.py", "--gpus", str(world), "--stub-renderer", "--steps", "2", "--warmup", "1",
                    "--batch", "3", "--repeats", "2"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == world and line["config"]["gather"]["check"] == "ok"
    assert len(line["per_rank_s"]["all"]) == world and line["repeats"] == 2
    assert abs(line["ms_per_step"] * 2 * line["value"] / 1e3 - 2 * 3 * world) < 0.05


def test_outer_torchrun_and_no_gather():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), "bench.py", "--gpus", "2", "--stub-renderer", "--steps", "2", "--warmup", "0",
           "--batch", "3", "--no-gather", "--sync-steps"]
    p, line = _run(cmd)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["config"]["distributed"]["launcher"].startswith("external")
    assert line["config"]["gather"]["mode"].startswith("off")


def test_world_size_mismatch_is_refused():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), "bench.py", "--gpus", "4", "--stub-renderer", "--steps", "1", "--warmup", "0"]
    p, line = _run(cmd)
    assert p.returncode != 0 and line is None
    assert "WORLD_SIZE=2 but --gpus 4" in p.stderr


def test_more_ranks_than_devices_fails_loudly():
    """The container has no HIP device: `--gpus 2` of the real renderer must refuse instead of printing an N = 1 line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs fewer than 2 devices")
    p, line = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 2 and line is None
    assert "HIP device(s) visible" in p.stderr


def test_single_rank_stub_has_no_gather():
    p, line = _run([sys.executable, "bench.py", "--stub-renderer", "--steps", "2", "--warmup", "1", "--batch", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["config"]["gather"]["mode"].startswith("off")
    assert line["config"]["distributed"]["launcher"] == "single process"


import pytest


@pytest.mark.gpu
def test_real_renderer_two_ranks_on_one_device_rehearsal(gpu_device):
    """The N > 1 code path with the REAL renderer: two ranks share the box's one GPU (gloo gather through host memory --
    RCCL refuses two ranks on one device), small scene.  Checks the launcher, per-rank sharding, pack + gather + the
    received-bytes check, and that the line is flagged as a rehearsal."""
    p, line = _run([sys.executable, "bench.py", "--gpus", "2", "--share-devices", "--backend", "gloo", "--scale", "0.03",
                    "--steps", "2", "--warmup", "1", "--batch", "4", "--views", "8", "--slots", "2", "--no-drop-in"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and "rehearsal" in line and line["config"]["distributed"]["backend"] == "gloo"
    g = line["config"]["gather"]
    assert g["check"] == "ok" and g["bytes_per_rank_and_batch"] == 4 * 800 * 800 * (3 + 2 + 1)
    assert g["views_per_s_render_only"] > g["views_per_s_with_gather"] > 0
    assert [d["device"] for d in line["config"]["distributed"]["devices"]] == ["cuda:0", "cuda:0"]
    assert line["roofline"]["kernel"] in ("composite", "preprocess", "bin_count", "bin_scatter", "tile_sort")


@pytest.mark.gpu
def test_dynamic_sequence_bench_line(gpu_device):
    """bench.py --dynamic (configs[4] as SURVEY.md section 8d writes it: poses from the reference's trajectory fixture) at a
    small scale: runs, names the sequence, and carries the posed CPU baseline."""
    p, line = _run([sys.executable, "bench.py", "--workload", "c5", "--dynamic", "--scale", "0.02", "--steps", "2", "--warmup", "1",
                    "--batch", "4", "--views", "200", "--no-drop-in", "--cpu-budget-s", "2"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "simulation_steps.json" in line["config"]["sequence"] and line["config"]["objects"] == 20
    assert line["cpu_baseline"]["value"] > 0 and line["value"] > 0


@pytest.mark.gpu
def test_one_rank_process_group_on_rccl(gpu_device):
    """What can be rehearsed of the RCCL path on a one-GPU box: a ONE-rank process group on the real backend (`--force-dist`):
    RCCL initialisation in this environment, all_gather_object / barrier / all_reduce of the timing protocol, the asynchronous
    gather of uint8 and 16-bit-as-bytes frames and its check -- every collective call the N-rank run makes, with N = 1."""
    p, line = _run([sys.executable, "bench.py", "--force-dist", "--backend", "nccl", "--scale", "0.03", "--steps", "2", "--warmup", "1",
                    "--batch", "4", "--views", "8", "--slots", "2", "--no-drop-in", "--no-cpu-baseline"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = line["config"]["distributed"]
    assert d["backend"] == "nccl (RCCL)" and d["world_size"] == 1 and line["n_gpus"] == 1 and "rehearsal" in line
    g = line["config"]["gather"]
    assert g["check"] == "ok" and g["bytes_per_rank_and_batch"] == 4 * 800 * 800 * (3 + 2 + 1)
    assert g["views_per_s_with_gather"] > 0 and g["views_per_s_render_only"] > 0


@pytest.mark.gpu
def test_dynamic_sequence_two_ranks_rehearsal(gpu_device):
    """configs[4]'s multi-GPU form, rehearsed: time step s -> rank s mod 2, every rank poses its own steps from the trajectory
    fixture, packs records and gathers them (two ranks on the one device, gloo).  Launcher, sharding of TIME STEPS, gather
    and check -- not a result."""
    p, line = _run([sys.executable, "bench.py", "--gpus", "2", "--share-devices", "--backend", "gloo", "--workload", "c5",
                    "--dynamic", "--scale", "0.02", "--steps", "2", "--warmup", "1", "--batch", "4", "--views", "200",
                    "--slots", "2", "--repeats", "2", "--no-drop-in"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and "rehearsal" in line and "simulation_steps.json" in line["config"]["sequence"]
    g = line["config"]["gather"]
    assert g["check"] == "ok" and g["views_per_s_with_gather"] > 0 and len(line["per_rank_s"]["all"]) == 2
    assert line["config"]["objects"] == 20


def test_cpu_only_line_for_config_0():
    """BASELINE.json configs[0] (10 k-Gaussian cube, 256x256, CPU rasterizer: plumbing, no GPU): `bench.py --cpu-only` times
    the oracle alone and says so; no HIP device is touched."""
    p, line = _run([sys.executable, "bench.py", "--cpu-only", "--cpu-budget-s", "1.5"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["cpu_only"] is True and line["n_gpus"] == 0 and line["value"] > 0
    assert "C1" in line["config"]["workload"] and line["cpu_baseline"]["kind"] == "port"
