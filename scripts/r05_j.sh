#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05j
rm -f ${P}_*
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > ${P}_pytest_gpu.txt
AB_TAGS="base w8" bash scripts/ab_libs.sh c3 > ${P}_ab_w8_c3.txt 2>&1
AB_TAGS="base w8" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_w8_c5.txt 2>&1
tail -8 ${P}_pytest_gpu.txt; cat ${P}_ab_w8_c3.txt ${P}_ab_w8_c5.txt
