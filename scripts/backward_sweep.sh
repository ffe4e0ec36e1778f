#!/bin/bash
cd "$(dirname "$0")/.."
cp pegasus_amd/csrc/libpegasus_raster.so /tmp/lib_orig.so
for f in build_variants/*.so; do
  cp "$f" pegasus_amd/csrc/libpegasus_raster.so
  echo -n "$f  "; python scripts/backward_bench.py c3 1.0 2>&1 | grep "forward+backward"
done
cp /tmp/lib_orig.so pegasus_amd/csrc/libpegasus_raster.so
