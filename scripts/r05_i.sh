#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05i
rm -f ${P}_*
for v in issue; do
  echo "== parity with lib_$v" >> ${P}_variant_parity.txt
  PGR_LIB=$PWD/build_variants/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "long_tile_lists or tie_index or very_long or c3_merged or full_size_view_matches or c5_view or grazing_views_match or fuzz or c1_cube or c2_object" 2>&1 | tail -3 >> ${P}_variant_parity.txt
done
AB_TAGS="base issue" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_c5.txt 2>&1
AB_TAGS="base issue" bash scripts/ab_libs.sh c3 > ${P}_ab_c3.txt 2>&1
cat ${P}_variant_parity.txt ${P}_ab_c5.txt ${P}_ab_c3.txt
