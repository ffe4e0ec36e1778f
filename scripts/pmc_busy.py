"""Per-kernel unit utilisation from the SQ counter summary scripts/pmc_run.sh writes (profiles/r02_pmc_sq_counters.txt).

    valu_busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles -> cycles) / 1024 SIMDs / kernel cycles
    lds_busy  = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles          kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(units: MI355X_MICROARCH.md "Per-instruction cycle constants": SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is
summed over the 8 XCDs -- cross-checked against the HIP-event kernel time: 2.50 M cycles for the 1.22 ms raster-only
compositor launch = 2.05 GHz under the profiler).  Usage: python3 scripts/pmc_busy.py <summary.txt> <out.json>"""
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(dict)
kernel = None
for line in open(sys.argv[1]):
    m = re.match(r"(pgr::\S.*?)\s+dispatches=(\d+)", line)
    if m:
        kernel = m.group(1)
        continue
    if "dispatches=" in line:        # some other kernel's block: its counters are not ours
        kernel = None
        continue
    m = re.match(r"\s+(\w+)\s+\d+\s+per-dispatch\s+(\d+)", line)
    if m and kernel:
        acc[kernel][m.group(1)] = float(m.group(2))
out = {}
for k, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_ACTIVE_INST_VALU" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = dict(kernel_cycles=round(cyc), valu_busy=round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4),
                  lds_busy=round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256 / cyc, 4),
                  lds_bank_conflict_share=round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 4),
                  valu_insts_per_simd_cycle=round(c.get("SQ_INSTS_VALU", 0.0) / 1024 / cyc, 4),
                  cycles_per_valu_inst=round(c["SQ_ACTIVE_INST_VALU"] * 4 / max(c.get("SQ_INSTS_VALU", 1.0), 1.0), 3))
json.dump(dict(source="rocprofv3 --pmc passes of scripts/pmc_run.sh (python3 bench.py --steps 2 --warmup 1 --batch 16 "
                      "--views 32 --raster-only --no-cpu-baseline --sync-steps), one counter group per pass",
               kernels=out), open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(f"{k:52s} valu {v['valu_busy']:.2f}  lds {v['lds_busy']:.2f}  conflicts {v['lds_bank_conflict_share']:.2f}  "
          f"{v['cycles_per_valu_inst']:.2f} cycles / VALU inst")
