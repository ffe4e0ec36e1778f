#!/usr/bin/env python3
"""The vector-issue model of bench.py's roofline (round 6): how many SIMD cycles a kernel's vector instructions NEED.

Three measured inputs, none of them a datasheet figure:
  1. profiles/r06_valu_classes.txt      issue cost of every instruction kind the hot loops use, in SHADER cycles per wave64
                                        instruction with 8 resident waves of independent work on every SIMD
                                        (scripts/microbench/valu_classes.hip; the shader clock of every row is in the file)
  2. profiles/r06_valu_classes_pmc.txt  which SQ instruction-class counter each kind is counted under (same binary under
                                        rocprofv3 --pmc): ADD_F32 = add / sub / pk_add, MUL_F32 = mul / pk_mul, FMA_F32 = fma /
                                        pk_fma (a packed instruction counts ONCE), TRANS_F32, CVT = v_cvt_*, INT32 = integer
                                        add / compare / bfe / mad / mul_lo / add3 / lshl_add / integer DPP; moves, selects, fp
                                        compares, min / max, shifts, logic, floor, lane reads fall under no class ("OTHER")
  3. profiles/pmc*.json                 the DYNAMIC count of every class per launch of every kernel of the bench command
                                        (scripts/pmc_profile.sh, class-counter passes) and the kernel's cycles
                                        (GRBM_GUI_ACTIVE / 8 XCDs) under the profiler

A class mixes 2.3-cycle and 4.2-cycle instructions (v_add_f32 vs v_pk_add_f32, v_mov vs v_cndmask), which the counters cannot
tell apart: the split INSIDE a class is taken from the kernel's ISA (hipcc --save-temps), every static instruction weighted
4^(loop depth).  Model:

    cycles_needed(kernel) = sum over classes  N_class (counter, per launch) x mean cost of the class in this kernel's ISA / 1024 SIMDs
    frac = cycles_needed / cycles the kernel took              (both in shader cycles: no clock assumption)

`frac` is what bench.py prints as roofline.frac for the dominant kernel (with the live clock from pgr_clock_probe turning the
live HIP-event duration into cycles).  Also printed: the same sum with every instruction at its class's cheapest / dearest
kind (the bracket the ISA split moves inside), and the scalar unit's share (SQ_INSTS_SALU / (256 CUs x cycles)).

    python3 scripts/issue_model.py [pmc.json ...]  ->  profiles/r06_issue_model.txt + profiles/issue_model.json
"""
import json
import re
import subprocess
import sys
import tempfile
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
N_SIMD, N_CU = 1024, 256


def kind_costs(path):
    """{mnemonic: shader cycles per wave-instruction at 8 waves per SIMD}"""
    cost = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+waves/SIMD 8 .*?([\d.]+) shader cycles per wave-instruction", line)
        if m:
            cost[m.group(1)] = float(m.group(2))
    return cost


def classify(mn, operands, cost):
    """(SQ class, cycles) of one ISA instruction.  Mnemonics are matched to the measured kinds of r06_valu_classes.txt; what
    was not measured takes the cost of its nearest measured relative (listed in the report)."""
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    dpp = mn.endswith("_dpp") or "row_" in operands or "quad_perm" in operands or mn.endswith("_sdwa")
    slow = cost["v_cndmask_b32_sgpr"]
    c = None
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", base):
        return "TRANS_F32", cost["v_exp_f32"] if "exp" in base else cost["v_rcp_f32"] if "rcp" in base else cost["v_sqrt_f32"]
    if base.startswith("v_pk_fma_f32"):
        return "FMA_F32", cost["v_pk_fma_f32"]
    if base.startswith("v_pk_mul_f32"):
        return "MUL_F32", cost["v_pk_mul_f32"]
    if base.startswith("v_pk_add_f32"):
        return "ADD_F32", cost["v_pk_add_f32"]
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mad)_f32", base):
        cls, c = "FMA_F32", cost["v_fma_f32"]
    elif re.match(r"v_mul(_legacy)?_f32", base):
        cls, c = "MUL_F32", cost["v_mul_f32"]
    elif re.match(r"v_(add|sub|subrev)_f32", base):
        cls, c = "ADD_F32", cost["v_add_f32"]
    elif base.startswith("v_cvt_"):
        cls, c = "CVT", cost["v_cvt_f32_u32"]
    elif re.match(r"v_(add|sub|subrev)(_co)?_u32|v_(addc|subb|subbrev)_co_u32|v_(add|sub)_i32", base):
        cls, c = "INT32", cost["v_add_u32"]
    elif re.match(r"v_cmpx?_\w+_[ui](32|16)", base):
        cls, c = "INT32", cost["v_cmp_lt_u32_sgpr"]
    elif re.match(r"v_(bfe_[ui]32|mad_[ui]32_[ui]24|mul_lo_u32|mul_hi_[ui]32|mul_[ui]32_[ui]24|add3_u32|lshl_add_u32|add_lshl_u32|"
                  r"(min|max|med3|min3|max3)_[ui]32|mad_i32_i24|mbcnt_\w+|sad_u32|lshl_or_b32|and_or_b32|or3_b32|xad_u32|bcnt_u32_b32)", base):
        cls, c = "INT32", cost["v_add3_u32"]
    elif re.match(r"v_\w+_[uib]64|v_mad_[ui]64_[ui]32|v_cmpx?_\w+_[ui]64", base):
        cls, c = "INT64", slow
    elif base.startswith("v_mov_b32") or re.match(r"v_(and|or|xor|not|xnor)_b32", base):
        cls, c = "OTHER", cost["v_mov_b32"] if base.startswith("v_mov") else cost["v_and_b32"]
    elif re.match(r"v_(lshlrev|lshrrev|ashrrev)_[bi]32", base):
        cls, c = "OTHER", cost["v_lshlrev_b32"]
    elif re.match(r"v_cmpx?_\w+_f32|v_cmp_class", base):
        cls, c = "OTHER", cost["v_cmp_lt_f32_sgpr"]
    elif re.match(r"v_(min|max|med3|min3|max3)_f32", base):
        cls, c = "OTHER", cost["v_min_f32"]
    elif re.match(r"v_(floor|ceil|trunc|rndne|fract)_f32", base):
        cls, c = "OTHER", cost["v_floor_f32"]
    elif re.match(r"v_(readlane|readfirstlane|writelane)_b32", base):
        cls, c = "OTHER", cost["v_readlane_b32"]
    elif base.startswith("v_cndmask"):
        cls, c = "OTHER", cost["v_cndmask_b32_sgpr"]
    else:
        cls, c = "OTHER", slow
    if dpp:
        c = max(c, cost["v_max_u32_dpp"])
    return cls, c


def kernel_isa(asm_text):
    """{short kernel name: [(mnemonic, operands, loop depth)]} of every VALU instruction."""
    out, cur, depth = {}, None, 0
    pending = None
    for line in asm_text.splitlines():
        m = re.match(r"^(_ZN3pgr\w+):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void pgr::", "").replace("pgr::", "")
            cur = out.setdefault(name, [])
            depth = 0
            continue
        if cur is None:
            continue
        if re.match(r"^\s*\.(amdhsa_kernel|Lfunc_end)", line):
            cur = None
            continue
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
            ds = [int(d) for d in re.findall(r"Depth=(\d+)", line)]
            depth = max(ds) if ds and ("in Loop" in line or "This" in line or "Parent" in line) else 0
            pending = True
            continue
        if pending and re.match(r"^\s+;", line):
            ds = [int(d) for d in re.findall(r"Depth=(\d+)", line)]
            if ds:
                depth = max(depth, max(ds))
            continue
        pending = False
        m = re.match(r"^\s+(v_\w+)\s*(.*)", line)
        if m and not m.group(1).startswith(("v_mfma", "v_accvgpr", "v_nop")):
            cur.append((m.group(1), m.group(2), depth))
    return out


def main(pmc_files):
    cost = kind_costs(ROOT / "profiles" / "r06_valu_classes.txt")
    from pegasus_amd.build import FLAGS
    with tempfile.TemporaryDirectory() as td:
        import os
        extra = os.environ.get("ISSUE_MODEL_HIPCC_FLAGS", "").split()      # (e.g. the -D switches of the build that was profiled)
        subprocess.run(["hipcc", *FLAGS, *extra, "--save-temps", "-o", f"{td}/x.so", str(ROOT / "pegasus_amd/csrc/pegasus_raster.hip")],
                       cwd=td, check=True, capture_output=True)
        asm = next(Path(td).glob("*gfx950*.s")).read_text()
    isa = kernel_isa(asm)
    lines = [l.rstrip() for l in __doc__.strip().splitlines()[:1]]
    lines += ["", "# issue cost by kind (shader cycles per wave64 instruction, 8 waves per SIMD; profiles/r06_valu_classes.txt):",
              "#   " + "  ".join(f"{k}={v:.2f}" for k, v in cost.items())]
    report = {"costs": cost, "kernels": {}}
    for pf in pmc_files:
        pmc = json.load(open(pf))
        lines += ["", f"## {Path(pf).name}: {pmc.get('workload')} ({pmc.get('source', '')[:0]}library {pmc.get('library_sha16')})",
                  f"{'kernel':56s} {'VALU inst':>11s} {'needed cyc':>11s} {'kernel cyc':>11s} {'frac':>6s} {'[cheapest':>10s} {'dearest]':>9s} "
                  f"{'cyc/inst':>8s} {'SALU':>6s}   class: share of the instructions x mean cycles in this kernel's ISA"]
        rows = []
        for kname, e in pmc["kernels"].items():
            cls_n = e.get("valu_classes_per_launch")
            if not cls_n or not e.get("kernel_cycles") or e.get("valu_insts_per_launch", 0) < 1e6:
                continue
            stat = isa.get(kname) or isa.get(re.sub(r"<.*", "", kname)) or []
            key = next((k for k in isa if k == kname or k.replace(" ", "") == kname.replace(" ", "")), None)
            stat = isa.get(key, stat)
            wsum, csum, lo, hi = defaultdict(float), defaultdict(float), {}, {}
            for mn, ops, depth in stat:
                c, cy = classify(mn, ops, cost)
                w = 4.0 ** depth
                wsum[c] += w
                csum[c] += w * cy
                lo[c] = min(lo.get(c, cy), cy)
                hi[c] = max(hi.get(c, cy), cy)
            default = cost["v_cndmask_b32_sgpr"]
            total = sum(cls_n.values())
            need = need_lo = need_hi = 0.0
            parts = []
            mean_cost = {}
            for c, n in cls_n.items():
                mean = csum[c] / wsum[c] if wsum[c] else default
                mean_cost[c] = mean
                need += n * mean
                need_lo += n * lo.get(c, mean)
                need_hi += n * hi.get(c, mean)
                if n / total >= 0.02:
                    parts.append(f"{c} {n / total:.2f} x {mean:.2f}")
            cyc = float(e["kernel_cycles"])
            salu = e["counters_per_launch"].get("SQ_INSTS_SALU", 0.0) / (N_CU * cyc)
            row = dict(valu_insts=total, cycles_needed=need / N_SIMD, kernel_cycles=cyc, frac=need / N_SIMD / cyc,
                       frac_cheapest=need_lo / N_SIMD / cyc, frac_dearest=need_hi / N_SIMD / cyc, cycles_per_inst=need / total,
                       salu_frac=salu, class_cost=mean_cost, static_valu_insts=len(stat))
            rows.append((kname, row, parts))
        for kname, row, parts in sorted(rows, key=lambda r: -r[1]["cycles_needed"]):
            lines.append(f"{kname:56s} {row['valu_insts']:11.4g} {row['cycles_needed']:11.4g} {row['kernel_cycles']:11.4g} {row['frac']:6.3f} "
                         f"{row['frac_cheapest']:10.3f} {row['frac_dearest']:9.3f} {row['cycles_per_inst']:8.2f} {row['salu_frac']:6.3f}   "
                         + ", ".join(parts))
            report["kernels"].setdefault(pmc.get("workload", Path(pf).stem), {})[kname] = {k: (round(v, 4) if isinstance(v, float) else v)
                                                                      for k, v in row.items() if k != "class_cost"}
            report["kernels"][pmc.get("workload", Path(pf).stem)][kname]["class_cost"] = {c: round(v, 3) for c, v in row["class_cost"].items()}
        report.setdefault("library_sha16", {})[pmc.get("workload", Path(pf).stem)] = pmc.get("library_sha16")
    lines += ["", "# frac = cycles the kernel's vector instructions need at the measured issue costs / cycles it took (GRBM_GUI_ACTIVE / 8, under",
              "# the profiler).  1.0 = the SIMDs' vector ports never idle.  The bracket is the same sum with every class at its cheapest / dearest",
              "# member found in the kernel.  SALU = scalar instructions per CU and cycle (one scalar issue per CU and cycle at most).",
              "# Not measured, priced like v_cndmask (4.2): 64-bit integer forms, v_perm / v_bfi / v_alignbit, SDWA forms."]
    text = "\n".join(lines) + "\n"
    (ROOT / "profiles" / "r06_issue_model.txt").write_text(text)
    (ROOT / "profiles" / "issue_model.json").write_text(json.dumps(report, indent=1) + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1:] or [str(ROOT / "profiles" / "r06_pmc.json"), str(ROOT / "profiles" / "r06_pmc_c5.json")])
