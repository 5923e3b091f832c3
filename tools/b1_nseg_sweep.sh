#!/bin/bash
# x segments of the fused x-march schedule at batch 1 (BASELINE config 2 as written): tune bits 16-23
cd "${GRAFT_REPO_ROOT:-.}"
base=$(( 3 | (2<<4) | (3<<8) | (1<<14) | (3<<24) | (2<<27) ))
for nseg in 0 2 3 4 5 8; do
  t=$(( base | (nseg<<16) ))
  python bench.py --batch-per-gpu 1 --steps 30 --warmup 5 --no-cpu-baseline --no-unet --tune $t 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nseg=$nseg', j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['value'])"
done
