#!/bin/bash
# rocprofv3 counter passes (one group of counters per run; no tracing flags) over a short raster-only bench.
#   scripts/pmc_run.sh <out-name> "<counters group 1>" "<counters group 2>" ...
# Writes gpurun_out/<out-name>.txt (per-kernel sums via scripts/pmc_summary.py).
cd "$(dirname "$0")/.."
REPO=$PWD
name=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/$name.txt
i=0
for grp in "$@"; do
  i=$((i+1))
  d=/tmp/pmc_${name}_$i
  rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --pmc $grp -d $d -- python3 $REPO/bench.py --steps 2 --warmup 1 --batch 16 --views 32 --raster-only --no-cpu-baseline --sync-steps > /tmp/pmc_${name}_$i.log 2>&1 )
  db=$(find $d -name "*.db" | head -1)
  echo "## counters: $grp" >> gpurun_out/$name.txt
  if [ -n "$db" ]; then python3 scripts/pmc_summary.py $db >> gpurun_out/$name.txt; else echo "no db; log tail:" >> gpurun_out/$name.txt; tail -5 /tmp/pmc_${name}_$i.log >> gpurun_out/$name.txt; fi
done
