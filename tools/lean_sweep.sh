#!/bin/bash
# tiles-per-block sweep of the lean few-channel kernel (NRT_LEAN_TPB), warp C = 1 / 3 at 4 x 160^3
cd "${GRAFT_REPO_ROOT:-.}"
for tpb in 1 2 4 8 16; do
  echo "tpb=$tpb"; NRT_LEAN_TPB=$tpb PYTHONPATH=. python tools/smallc_bench.py 2>/dev/null | grep -E '"C": (1|3), "method": "linear"|VecInt'
done
