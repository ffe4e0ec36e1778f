# A/B of two builds of the library on one box:  ab_libs.sh <workload> [bench flags]   (build_variants/lib_base.so, lib_new.so)
cd $GRAFT_REPO_ROOT
wl=${1:-c3}; shift
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
for rep in 1 2; do
for v in base new; do
  cp build_variants/lib_$v.so pegasus_amd/csrc/libpegasus_raster.so
  python bench.py --no-drop-in --no-cpu-baseline --workload $wl "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['stage_ms_per_view'])"
done
done
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
