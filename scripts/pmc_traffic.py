"""Per-stage HBM-side traffic of the frames path from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes).

Usage (GPU box):  python3 scripts/pmc_traffic.py <fetch.db> <write.db> <out.json> <out.txt>
Counter unit: KiB.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 the counter reports half the
bytes of 16-B-per-lane loads -- every bulk read of this pipeline is a global_load_dwordx4; calibration of this
project's own patterns: scripts/microbench/fetch_calib.hip, profiles/r01_m_fetch_calibration.txt)."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

STAGE_OF = [("preprocess_batch_kernel", "preprocess"), ("bin_kernel<false>", "bin_count"), ("bin_kernel<true>", "bin_scatter"),
            ("tile_sort", "tile_sort"), ("order_", "tile_sort"), ("composite_quarter_kernel<false, true>", "composite")]


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
    acc, disp = defaultdict(float), defaultdict(set)
    for r in cur.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        m = re.search(r"pgr::(\w+)(<[^>]*>)?", r[ix[name_col]])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "")
        acc[k] += float(r[ix["value"]])
        disp[k].add(r[ix["dispatch_id"]])
    return acc, disp


def main(fetch_db, write_db, out_json, out_txt, batch=32, workload="c3", command=""):
    f, fd = per_kernel(fetch_db, "FETCH_SIZE")
    w, wd = per_kernel(write_db, "WRITE_SIZE")
    n_batches = len(fd.get("preprocess_batch_kernel<3, false>", fd.get("preprocess_batch_kernel<3>", {1})))
    stages = defaultdict(lambda: dict(fetch_kib=0.0, write_kib=0.0))
    lines = [f"# rocprofv3 --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (separate pass), values in KiB, summed per kernel over "
             f"{n_batches} batches of {batch} views", f"# command: {command}",
             f"{'kernel':58s} {'dispatches':>10s} {'FETCH_SIZE KiB':>16s} {'WRITE_SIZE KiB':>16s}"]
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, 0) + w.get(k, 0))):
        lines.append(f"{k:58s} {len(fd.get(k, wd.get(k, []))):10d} {f.get(k, 0):16.0f} {w.get(k, 0):16.0f}")
        for pat, st in STAGE_OF:
            if k.startswith(pat) or pat in k:
                stages[st]["fetch_kib"] += f.get(k, 0) / n_batches
                stages[st]["write_kib"] += w.get(k, 0) / n_batches
                break
    out = {"workload": workload, "batch": batch,
           "source": ("profiles (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; " + command +
                      "; per batch (one compositor launch); counter unit KiB; FETCH_SIZE doubled: gfx950 correction for 16-B-per-lane loads, "
                      "MI355X_MICROARCH.md HBM section, confirmed on this pipeline's gather pattern by "
                      "scripts/microbench/fetch_calib.hip)"),
           "kernels": {st: {"fetch_bytes_per_launch": int(2 * v["fetch_kib"] * 1024), "write_bytes_per_launch": int(v["write_kib"] * 1024),
                            "fetch_size_kib_raw": round(v["fetch_kib"]), "write_size_kib_raw": round(v["write_kib"])}
                       for st, v in stages.items()}}
    json.dump(out, open(out_json, "w"), indent=1)
    lines.append("")
    lines.append(f"# per stage and {batch}-view batch (composite = the fused frames compositor): raw KiB, and bytes with FETCH_SIZE x 2")
    for st, v in out["kernels"].items():
        lines.append(f"{st:14s} fetch {v['fetch_size_kib_raw']:10d} KiB  write {v['write_size_kib_raw']:10d} KiB   "
                     f"-> {v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']:14d} B")
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:5], command=sys.argv[5] if len(sys.argv) > 5 else "")
