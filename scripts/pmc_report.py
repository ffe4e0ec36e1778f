"""One report from the rocprofv3 --pmc passes of scripts/pmc_profile.sh: per kernel, every counter summed over the kernel's
OWN dispatches and divided by that kernel's OWN dispatch count (round 2 divided everything by the preprocess kernel's count
and understated the compositor 1.43x), then

    traffic_bytes_per_launch = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024
        (counter unit KiB; FETCH_SIZE doubled: MI355X_MICROARCH.md "HBM": on gfx950 it reports half the bytes of 16-B-per-lane
         loads -- every bulk read of this pipeline is one; calibration on this pipeline's own patterns:
         scripts/microbench/fetch_calib.hip, profiles/r02_fetch_calibration.txt)
    valu_busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles -> cycles) / 1024 SIMDs / kernel cycles
    lds_busy  = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles            kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs

Usage: python3 scripts/pmc_report.py <out.json> <out.txt> "<command>" <pass1.db> <pass2.db> ..."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

STAGE_KERNEL = [("preprocess", r"preprocess_batch_kernel"), ("bin_count", r"bin_kernel<false"), ("bin_scatter", r"bin_kernel<true"),
                ("tile_sort", r"tile_sort_kernel$"), ("composite", r"composite_quarter_kernel")]


def short(name):
    m = re.search(r"pgr::(\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def read(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
    acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for r in cur.execute("select * from counters_collection"):
        k = short(r[ix[name_col]])
        if not k:
            continue
        acc[k][r[ix["counter_name"]]] += float(r[ix["value"]])
        disp[k].add(r[ix["dispatch_id"]])
    return acc, disp


def main(out_json, out_txt, command, *dbs):
    per = defaultdict(dict)          # kernel -> counter -> per-dispatch average
    ndisp = {}
    for p in dbs:
        if p == "MISSING":
            continue
        acc, disp = read(p)
        for k, cs in acc.items():
            n = len(disp[k])
            ndisp[k] = n if k not in ndisp else min(ndisp[k], n)
            for c, v in cs.items():
                per[k][c] = v / n
    kernels = {}
    for k, c in per.items():
        e = {"dispatches": ndisp[k]}
        if "FETCH_SIZE" in c:
            e["fetch_size_kib_per_launch"] = round(c["FETCH_SIZE"], 1)
        if "WRITE_SIZE" in c:
            e["write_size_kib_per_launch"] = round(c["WRITE_SIZE"], 1)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["traffic_bytes_per_launch"] = int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            e["kernel_cycles"] = round(cyc)
            e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4)
            if "SQ_INSTS_VALU" in c:
                e["cycles_per_valu_inst"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / max(c["SQ_INSTS_VALU"], 1.0), 3)
                e["valu_insts_per_launch"] = round(c["SQ_INSTS_VALU"])
            if "SQ_LDS_IDX_ACTIVE" in c:
                e["lds_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 4)
                e["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
        # the kernel's vector instructions by SQ class (per launch): what scripts/issue_model.py prices; "OTHER" = the rest of
        # SQ_INSTS_VALU (moves, selects, compares, min / max, lane ops, bit ops the class counters do not name)
        cls = {n[len("SQ_INSTS_VALU_"):]: c[n] for n in c if n.startswith("SQ_INSTS_VALU_") and "MFMA" not in n}
        if cls and "SQ_INSTS_VALU" in c:
            cls["OTHER"] = max(0.0, c["SQ_INSTS_VALU"] - sum(cls.values()))
            e["valu_classes_per_launch"] = {n: round(v) for n, v in sorted(cls.items())}
        e["counters_per_launch"] = {cc: round(v, 1) for cc, v in sorted(c.items())}
        kernels[k] = e
    stage_kernel = {}
    for st, pat in STAGE_KERNEL:
        cands = [k for k in kernels if re.search(pat, k)]
        if cands:       # the variant that ran most often in this command (the timed path's)
            stage_kernel[st] = max(cands, key=lambda k: (kernels[k]["dispatches"], kernels[k].get("traffic_bytes_per_launch", 0)))
    m = re.search(r"--batch (\d+)", command)
    w = re.search(r"--workload (\w+)", command)
    # which kernels the counters describe: the source hash compiled into the profiled library (pgr_version() ends with it;
    # bench.py compares it with the library it loaded -- a counter file of older kernels must not price today's durations)
    import ctypes
    import os
    lib_path = os.environ.get("PGR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pegasus_amd", "csrc",
                                                         "libpegasus_raster.so")
    try:
        h = ctypes.CDLL(lib_path)
        h.pgr_version.restype = ctypes.c_char_p
        lib_sha = h.pgr_version().decode().rsplit(" ", 1)[-1]
    except (OSError, AttributeError):
        lib_sha = None
    out = {"workload": w.group(1) if w else "c3", "batch": int(m.group(1)) if m else 32, "library_sha16": lib_sha,
           "fused": "--raster-only" not in command and "--separate-semantic" not in command,
           "source": ("rocprofv3 --pmc, one counter group per pass (FETCH_SIZE | WRITE_SIZE | SQ VALU group | SQ LDS group) over `" + command +
                      "` (scripts/pmc_profile.sh -> scripts/pmc_report.py); per kernel: counter sum / that kernel's own dispatch count; "
                      "unit KiB; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)"),
           "busy_definition": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles); SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); "
                              "kernel cycles = GRBM_GUI_ACTIVE / 8",
           "stage_kernel": stage_kernel, "kernels": kernels}
    json.dump(out, open(out_json, "w"), indent=1)
    lines = [f"# {out['source']}", "",
             f"{'kernel':52s} {'disp':>5s} {'FETCH KiB':>12s} {'WRITE KiB':>12s} {'traffic MB':>11s} {'VALU':>6s} {'LDS':>6s} {'confl':>6s} {'cyc/VALU':>8s}"]
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("traffic_bytes_per_launch", 0)):
        f = lambda key, fmt: (fmt % e[key]) if key in e else "-"
        lines.append(f"{k:52s} {e['dispatches']:5d} {f('fetch_size_kib_per_launch', '%.0f'):>12s} {f('write_size_kib_per_launch', '%.0f'):>12s} "
                     f"{('%.1f' % (e['traffic_bytes_per_launch'] / 1e6)) if 'traffic_bytes_per_launch' in e else '-':>11s} "
                     f"{f('valu_busy', '%.3f'):>6s} {f('lds_busy', '%.3f'):>6s} {f('lds_bank_conflict_share', '%.3f'):>6s} {f('cycles_per_valu_inst', '%.2f'):>8s}")
    lines += ["", "# per launch (= per dispatch of that kernel; the compositor and the binning kernels cover a whole batch per launch)",
              "# stage -> kernel: " + ", ".join(f"{s}={k}" for s, k in stage_kernel.items())]
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:])
