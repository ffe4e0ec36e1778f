# compositor work statistics (batch path, fused frames) from a -DPGR_COMP_STATS build in build_variants/lib_stats.so
cd $GRAFT_REPO_ROOT
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
cp build_variants/lib_stats.so pegasus_amd/csrc/libpegasus_raster.so
python scripts/comp_stats.py c3 frames 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
