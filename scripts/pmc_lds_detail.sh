#!/bin/bash
# One more rocprofv3 --pmc pass over the bench command: WHAT the LDS conflict cycles of the binning walks and the per-tile
# sorts are -- bank conflicts proper (SQ_LDS_BANK_CONFLICT), same-address serialisation (SQ_LDS_ADDR_CONFLICT), and how many
# of the LDS instructions are atomics -> gpurun_out/<name>.txt       scripts/pmc_lds_detail.sh <out-name> [bench flags]
cd "$(dirname "$0")/.."
REPO=$PWD
name=${1:-pmc_lds}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_FLAGS="--steps 16 --warmup 1 --repeats 1 --no-cpu-baseline --no-drop-in --profile-steps 1 --sync-steps $*"
d=/tmp/pmc_lds_$name
rm -rf $d
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE -d $d -- python3 $REPO/bench.py $BENCH_FLAGS > /tmp/pmc_lds_$name.log 2>&1 )
db=$(find $d -name "*.db" | head -1)
if [ -z "$db" ]; then echo "no database; log tail:"; tail -5 /tmp/pmc_lds_$name.log; exit 1; fi
python3 - "$db" "gpurun_out/$name.txt" "python3 bench.py $BENCH_FLAGS" <<'PY'
import sys
sys.path.insert(0, "scripts")
from pmc_report import read
acc, disp = read(sys.argv[1])
lines = [f"# rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN "
         f"SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE over `{sys.argv[3]}` (scripts/pmc_lds_detail.sh); per kernel: counter sum / its own dispatch count",
         "# bank = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; addr = SQ_LDS_ADDR_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles lost to lanes of one",
         "# instruction hitting the SAME word with an atomic); atomics = SQ_INSTS_LDS_ATOMIC / SQ_INSTS_LDS; lds busy = IDX_ACTIVE / (256 CUs x kernel cycles)",
         f"{'kernel':52s} {'disp':>5s} {'lds busy':>9s} {'bank':>7s} {'addr':>7s} {'atomics':>8s} {'LDS insts':>12s}"]
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_LDS_IDX_ACTIVE", 0.0) / max(len(disp[k]), 1)):
    c = {n: v / len(disp[k]) for n, v in acc[k].items()}
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if act <= 0:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    lines.append(f"{k:52s} {len(disp[k]):5d} {act / 256 / max(cyc, 1):9.3f} {c.get('SQ_LDS_BANK_CONFLICT', 0) / act:7.3f} "
                 f"{c.get('SQ_LDS_ADDR_CONFLICT', 0) / act:7.3f} {c.get('SQ_INSTS_LDS_ATOMIC', 0) / max(c.get('SQ_INSTS_LDS', 1), 1):8.3f} {c.get('SQ_INSTS_LDS', 0):12.0f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
