#!/bin/bash
# Compositor tuning sweep: builds libpegasus_raster.so variants (here, no GPU needed) or runs bench.py over them (GPU box).
#   scripts/variant_sweep.sh build "U:W U:W ..." [extra -D flags]   -> build_variants/lib_u<U>_w<W>.so
#     W = PGR_COMP_WAVES (compositor occupancy cap, 0 = compiler's choice); U is a label only (the unroll knob of the
#     half-tile compositor it once selected is gone) -- pass other knobs as extra -D flags
#   scripts/variant_sweep.sh run  [bench args]                      -> one result line per variant
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  specs=$1; shift
  for s in $specs; do
    U=${s%%:*}; Wv=${s##*:}
    ( cd pegasus_amd/csrc && hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics -fPIC -shared --offload-arch=gfx950 \
        -DPGR_COMP_WAVES=$Wv "$@" -o ../../build_variants/lib_u${U}_w${Wv}.so pegasus_raster.hip ) &
  done
  wait
  ls -la build_variants
else
  for f in build_variants/*.so; do      # (loaded through PGR_LIB: the product .so is never overwritten)
    echo -n "$f  "
    PGR_LIB=$PWD/$f timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); st = d['roofline']['stage_ms_per_view']
        print(round(d['value'], 1), d['unit'], {k: round(v, 4) for k, v in st.items()})
"
  done
fi
