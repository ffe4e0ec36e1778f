#!/bin/bash
# work statistics from a -DPGR_COMP_STATS / -DPGR_SORT_STATS build in build_variants/lib_stats.so:  r03_stats.sh <workload> [frames]
# (the variant library is loaded through PGR_LIB; the product .so is never touched)
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}"
PGR_LIB=$PWD/build_variants/lib_stats.so python scripts/comp_stats.py ${1:-c3} ${2:-} 2>&1 | grep -v amdgpu.ids
