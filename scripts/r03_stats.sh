# work statistics from a -DPGR_COMP_STATS / -DPGR_SORT_STATS build in build_variants/lib_stats.so:  r03_stats.sh <workload> [frames]
cd $GRAFT_REPO_ROOT
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
cp build_variants/lib_stats.so pegasus_amd/csrc/libpegasus_raster.so
python scripts/comp_stats.py ${1:-c3} ${2:-} 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
