#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over a short frames bench + the calibration microbenchmark -> gpurun_out/
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --sync-steps"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$c -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --sync-steps > /tmp/pmc_$c.log 2>&1 )
done
python3 scripts/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) \
    gpurun_out/pmc_traffic.json gpurun_out/pmc_fetch_write.txt "$CMD"
rm -rf /tmp/pmc_calib
( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_calib -- $REPO/scripts/microbench/fetch_calib > /tmp/pmc_calib.log 2>&1 )
{ echo "# scripts/microbench/fetch_calib under rocprofv3 --pmc FETCH_SIZE (KiB)"; grep expected /tmp/pmc_calib.log;
  python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/pmc_calib/**/*.db", recursive=True)[0])
cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
ix = {c: i for i, c in enumerate(cols)}
name = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
for r in db.execute("select * from counters_collection order by dispatch_id"):
    print(f"{r[ix[name]][:40]:40s} dispatch {r[ix['dispatch_id']]}  {r[ix['counter_name']]} = {r[ix['value']]:.0f} KiB = {r[ix['value']]*1024:.4g} B")
PY
} > gpurun_out/fetch_calibration.txt
cat gpurun_out/fetch_calibration.txt
