cd $GRAFT_REPO_ROOT
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
cp build_variants/lib_sorttiming.so pegasus_amd/csrc/libpegasus_raster.so
python scripts/sort_timing.py c5 2>&1 | grep -v amdgpu.ids
python scripts/sort_timing.py c3 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
