cd $GRAFT_REPO_ROOT
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
for f in build_variants/*.so; do
  cp "$f" pegasus_amd/csrc/libpegasus_raster.so
  echo "== $f"
  python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
  python scripts/single_view_calls.py 64 c3 2>&1 | tail -1
done
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
bash scripts/single_view_trace.sh r03_d 40 c3 > /dev/null 2>&1
tail -32 gpurun_out/r03_d_single_view_timeline.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
