#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05h
rm -f ${P}_*
for v in t1c t1d; do
  echo "== parity with lib_$v" >> ${P}_variant_parity.txt
  PGR_LIB=$PWD/build_variants/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "long_tile_lists or tie_index or very_long or c3_merged or full_size_view_matches or c5_view or grazing_views_match" 2>&1 | tail -3 >> ${P}_variant_parity.txt
done
AB_TAGS="base t1c t1d" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_t1_c5.txt 2>&1
AB_TAGS="base t1c t1d" bash scripts/ab_libs.sh c3 > ${P}_ab_t1_c3.txt 2>&1
timeout 600 python -m pytest tests/test_facade_gpu.py -m gpu -q 2>&1 | tail -3 > ${P}_facade_tests.txt
bash scripts/single_view_trace.sh r05h 40 c3 > /dev/null 2>&1
cat ${P}_variant_parity.txt ${P}_ab_t1_c5.txt ${P}_ab_t1_c3.txt ${P}_facade_tests.txt; tail -24 gpurun_out/r05h_single_view_timeline.txt
