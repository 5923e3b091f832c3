#!/bin/bash
# (y, z) patch shape of the fused x-march schedule: does a longer z extent (more L1 sets: the row index mod 64 only varies with
# z and the parity of y at 160 x 160 planes) raise the L1 hit rate?  tune = ltx | lty<<4 | ltz<<8 | 1<<14 | lry<<24 | lrz<<27
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  t=$(( 3 | ($1<<4) | ($2<<8) | (1<<14) | ($3<<24) | ($4<<27) ))
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-unet --no-batch1 --tune $t 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lty=$1 ltz=$2 lry=$3 lrz=$4', j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
}
run 2 3 3 2
run 1 4 4 1
run 0 5 5 0
run 3 2 2 3
run 1 4 3 1
run 1 4 4 2
run 0 5 4 0
run 0 5 5 1
