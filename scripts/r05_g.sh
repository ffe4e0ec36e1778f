#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05g
rm -f ${P}_*
for v in s768 s1024 s1536; do
  echo "== parity with lib_$v" >> ${P}_variant_parity.txt
  PGR_LIB=$PWD/build_variants/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "long_tile_lists or tie_index or very_long or c3_merged or full_size_view_matches or c1_cube or c2_object or fuzz" 2>&1 | tail -3 >> ${P}_variant_parity.txt
done
AB_TAGS="base s768 s1024 s1536" bash scripts/ab_libs.sh c3 > ${P}_ab_short_c3.txt 2>&1
AB_TAGS="base s768 s1024 s1536" bash scripts/ab_libs.sh c2 --views 64 > ${P}_ab_short_c2.txt 2>&1
cat ${P}_variant_parity.txt ${P}_ab_short_c3.txt ${P}_ab_short_c2.txt
