#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05f
for v in t2a t2b t1a t1b; do
  echo "== parity with lib_$v" >> ${P}_variant_parity.txt
  PGR_LIB=$PWD/build_variants/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "long_tile_lists or tie_index or very_long or c5_view or c3_merged or full_size_view_matches or grazing_views_match" 2>&1 | tail -3 >> ${P}_variant_parity.txt
done
AB_TAGS="base t2a t2b t1a t1b" bash scripts/ab_libs.sh c3 > ${P}_ab_tiers_c3.txt 2>&1
AB_TAGS="base t2a t2b t1a t1b" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_tiers_c5.txt 2>&1
cat ${P}_variant_parity.txt ${P}_ab_tiers_c3.txt ${P}_ab_tiers_c5.txt
