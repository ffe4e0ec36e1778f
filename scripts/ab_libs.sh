#!/bin/bash
# A/B of builds of the library on one box:  ab_libs.sh <workload> [bench flags]
# Every build_variants/lib_<tag>.so named in $AB_TAGS (default "base new") is benched twice, interleaved; the library is
# chosen through PGR_LIB (pegasus_amd/_lib.py), the product .so is never overwritten.
set -e
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set}"
wl=${1:-c3}; shift || true
for rep in 1 2; do
for v in ${AB_TAGS:-base new}; do
  PGR_LIB=$PWD/build_variants/lib_$v.so python bench.py --no-drop-in --no-cpu-baseline --workload $wl "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['value_min'], d['value_max'], d['roofline']['stage_ms_per_view'])"
done
done
