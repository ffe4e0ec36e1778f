#!/bin/bash
# A/B of compile-time variants of the library on one GPU box, interleaved, the product .so untouched:
#   scripts/ab_variants.sh build "<tag>[:<hipcc -D flags>] ..."     (here, no GPU needed) -> build_variants/lib_<tag>.so
#   scripts/ab_variants.sh run   "<tag> ..." <workload> [bench flags] (GPU box)          -> one line per variant and repeat
#   scripts/ab_variants.sh parity "<tag> ..." "<pytest -k expression>" (GPU box)         -> the list-parity tests under each variant
# Round 5's A/B files (profiles/r05_sort_ab.txt, r05_sort_tier_shapes_ab.txt) were produced this way from variant flags that
# existed in the source at the time (PGR_SORT_POS, PGR_T1_SHAPE / PGR_T2_SHAPE, PGR_SHORT_NB, PGR_SORT_EH, PGR_COMP_WAVES); the
# winners are the code, the flags of the losers are gone with them (history: git log -p pegasus_amd/csrc/tilebin.hip.h).
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
FL="-O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics -fPIC -shared --offload-arch=gfx950"
case "$mode" in
  build)
    mkdir -p build_variants
    for v in $1; do
      tag=${v%%:*}; fl=""; [ "$v" != "$tag" ] && fl=${v#*:}
      ( cd pegasus_amd/csrc && hipcc $FL $fl -o ../../build_variants/lib_$tag.so pegasus_raster.hip ) &
    done
    wait; ls -la build_variants ;;
  run)
    tags=$1; wl=$2; shift 2
    AB_TAGS="$tags" bash scripts/ab_libs.sh $wl "$@" ;;
  parity)
    for v in $1; do
      echo "== parity with lib_$v"
      PGR_LIB=$PWD/build_variants/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "$2" 2>&1 | tail -3
    done ;;
  *) echo "usage: see the header of $0" >&2; exit 2 ;;
esac
