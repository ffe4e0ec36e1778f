#!/bin/bash
# round 6: row-coded candidate enumeration of the binning walks (preprocess.hip.h row_code) against the box enumeration
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_facade_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r06_rows_suite.txt
{
echo "# base = -DPGR_ROW_CODES=0 (every splat enumerates its candidate box, rounds 1-5), rows = per-row intervals"
echo "# C3 (default)"
bash scripts/ab_variants.sh run "base rows" c3
echo "# C5 (--workload c5 --views 200)"
bash scripts/ab_variants.sh run "base rows" c5 --views 200
} > gpurun_out/r06_rows_ab.txt 2>&1
export PGR_LIB=$PWD/build_variants/lib_rows.so
bash scripts/trace_run.sh r06_c5_rows --no-drop-in --sync-steps --workload c5 --views 200 > /dev/null 2>&1
