#!/bin/bash
# forward+backward time of every build in build_variants/ (loaded through PGR_LIB: the product .so is never overwritten)
cd "$(dirname "$0")/.."
for f in build_variants/*.so; do
  echo -n "$f  "; PGR_LIB=$PWD/$f python scripts/backward_bench.py c3 1.0 2>&1 | grep "forward+backward"
done
