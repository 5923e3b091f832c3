#!/bin/bash
# tile geometry of the lean tile kernel (variant 8): log2 z / y extents of the 256-voxel tile (x takes the rest); a wave is 64
# consecutive voxels of the tile (z fastest)
cd "$(dirname "$0")/.."
export PYTHONPATH=.
for geo in 82 81 66 65 67 50 51 52 34 35 36; do
  echo "geo ltz=$((geo>>4)) lty=$((geo&15))"
  NRT_LEAN_GEO=$geo timeout 120 python tools/smallc_bench.py 2>/dev/null | grep '"linear"' | head -4 | cut -c1-90
done
