#!/bin/bash
# Round-5 run B: full GPU suite (position-owned sort in the product build), then A/B of the two sort builds on C3 and C5
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05b
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > ${P}_pytest_gpu.txt
AB_TAGS="base new" bash scripts/ab_libs.sh c3 > ${P}_ab_sort_c3.txt 2>&1
AB_TAGS="base new" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_sort_c5.txt 2>&1
cat ${P}_pytest_gpu.txt ${P}_ab_sort_c3.txt ${P}_ab_sort_c5.txt
