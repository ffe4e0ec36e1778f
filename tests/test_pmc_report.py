"""scripts/pmc_report.py turns rocprofv3 --pmc databases into the per-kernel, per-launch numbers bench.py reports as
roofline.traffic / valu_issue.  Round 2's script divided every kernel's counter sum by the PREPROCESS kernel's dispatch count
and understated the compositor 1.43x; this pins the rule on a synthetic database: each kernel is divided
by its OWN dispatch count, FETCH_SIZE is doubled, units are KiB."""
import json
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _db(path, rows):
    db = sqlite3.connect(path)
    db.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real)")
    db.executemany("insert into counters_collection values (?,?,?,?)", rows)
    db.commit()
    db.close()


def test_each_kernel_is_divided_by_its_own_dispatch_count(tmp_path):
    comp = "void pgr::composite_quarter_kernel<false, true>(pgr::ViewEntry const*, unsigned int, unsigned int const*, pgr::SemanticDev)"
    pre = "void pgr::preprocess_batch_kernel<3, false>(PgrScene, pgr::CameraDev const*, pgr::PreOut const*, int, pgr::PosedDev, unsigned int*, int)"
    other = "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>>(int, at::native::FillFunctor<float>)"
    fetch, write, sq = [], [], []
    for d in range(10):                                    # the preprocess kernel ran 10 times ...
        fetch.append((d, pre, "FETCH_SIZE", 100.0)); write.append((d, pre, "WRITE_SIZE", 50.0))
        sq += [(d, pre, "SQ_ACTIVE_INST_VALU", 1024 * 1000 / 4 * 0.5), (d, pre, "GRBM_GUI_ACTIVE", 8 * 1000.0)]
    for d in range(100, 107):                              # ... the compositor 7 times
        fetch.append((d, comp, "FETCH_SIZE", 2000.0)); write.append((d, comp, "WRITE_SIZE", 700.0))
        sq += [(d, comp, "SQ_ACTIVE_INST_VALU", 1024 * 2000 / 4 * 0.8), (d, comp, "GRBM_GUI_ACTIVE", 8 * 2000.0),
               (d, comp, "SQ_INSTS_VALU", 1.0e6)]
    fetch.append((500, other, "FETCH_SIZE", 1.0e9))         # not one of ours: ignored
    paths = [str(tmp_path / f"{n}.db") for n in ("fetch", "write", "sq")]
    for p, rows in zip(paths, (fetch, write, sq)):
        _db(p, rows)
    out_json, out_txt = str(tmp_path / "pmc.json"), str(tmp_path / "pmc.txt")
    cmd = "python3 bench.py --steps 3 --warmup 1 --sync-steps"
    subprocess.run([sys.executable, str(ROOT / "scripts" / "pmc_report.py"), out_json, out_txt, cmd, *paths, "MISSING"], check=True,
                   capture_output=True)
    d = json.loads(Path(out_json).read_text())
    assert d["workload"] == "c3" and d["batch"] == 32 and d["fused"] is True
    k = d["kernels"]["composite_quarter_kernel<false, true>"]
    assert k["dispatches"] == 7
    assert k["fetch_size_kib_per_launch"] == 2000.0 and k["write_size_kib_per_launch"] == 700.0
    assert k["traffic_bytes_per_launch"] == int(2 * 2000 * 1024 + 700 * 1024)
    assert abs(k["valu_busy"] - 0.8) < 1e-4 and k["kernel_cycles"] == 2000
    p = d["kernels"]["preprocess_batch_kernel<3, false>"]
    assert p["dispatches"] == 10 and p["traffic_bytes_per_launch"] == int(2 * 100 * 1024 + 50 * 1024) and abs(p["valu_busy"] - 0.5) < 1e-4
    assert d["stage_kernel"]["composite"] == "composite_quarter_kernel<false, true>"
    assert d["stage_kernel"]["preprocess"] == "preprocess_batch_kernel<3, false>"
    assert not any("vectorized" in name for name in d["kernels"])
    assert "composite_quarter_kernel<false, true>" in Path(out_txt).read_text()
