#!/bin/bash
# per-tier effect of the sort's in-flight LDS batching: kernel-trace averages of the sort kernels per variant (C5, --sync-steps)
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
P=gpurun_out/r05k
rm -f ${P}_*
for v in base eh2 eh4 eh8; do
  PGR_LIB=$PWD/build_variants/lib_$v.so bash scripts/trace_run.sh r05k_$v --no-drop-in --sync-steps --workload c5 --views 200 --repeats 2 > /dev/null 2>&1
  echo "== $v" >> ${P}_tiers.txt; grep "tile_sort" gpurun_out/r05k_${v}_kernel_stats.txt >> ${P}_tiers.txt
done
AB_TAGS="base eh2 eh4 eh8" bash scripts/ab_libs.sh c5 --views 200 > ${P}_ab_c5.txt 2>&1
cat ${P}_tiers.txt ${P}_ab_c5.txt
