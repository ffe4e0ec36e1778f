#!/bin/bash
# round 6: the windowed sort (8193..15872 keys) and the split pre-pass (longer lists -> depth segments for the 512 x 16 kernel)
# against round 5's open-ended kernel, on one box.  Variants: scripts/ab_variants.sh build (see profiles/r06_sort_ab.txt)
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
tags=${1:-"base win full"}
{
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sort or long_lists or windowed or tie_index or records" 2>&1 | tail -3
python -m pytest tests/test_contract_edges.py tests/test_reference_pinned.py -q -m gpu 2>&1 | tail -3
} > gpurun_out/r06_sort_parity.txt 2>&1
{
echo "# C5 (--workload c5 --views 200): base = every list > 8192 keys in the open-ended kernel (round 5); win = + windowed sort (8193..15872 keys); full = + split pre-pass (> 15872 keys)"
bash scripts/ab_variants.sh run "$tags" c5 --views 200
echo "# C3 (default)"
bash scripts/ab_variants.sh run "$tags" c3
} > gpurun_out/r06_sort_ab.txt 2>&1
export PGR_LIB=$PWD/build_variants/lib_full.so
bash scripts/trace_run.sh r06_c5_full --no-drop-in --sync-steps --workload c5 --views 200 > /dev/null 2>&1
bash scripts/trace_run.sh r06_c3_full --no-drop-in --sync-steps > /dev/null 2>&1
