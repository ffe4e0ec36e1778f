#!/bin/bash
# per-phase sort timing from a -DPGR_SORT_TIMING build in build_variants/lib_sorttiming.so (loaded through PGR_LIB)
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}"
PGR_LIB=$PWD/build_variants/lib_sorttiming.so python scripts/sort_timing.py c5 2>&1 | grep -v amdgpu.ids
PGR_LIB=$PWD/build_variants/lib_sorttiming.so python scripts/sort_timing.py c3 2>&1 | grep -v amdgpu.ids
