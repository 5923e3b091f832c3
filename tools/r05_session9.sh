#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
FUSED_VARIANTS="BASE=0 STAGGER=1 STAGGER=2 STAGGER=4" FUSED_REPS=2 FUSED_STEPS=60 python tools/fused_variants.py 2>&1 | tee gpurun_out/s9_variants.jsonl
for v in BASE0 STAGGER1 STAGGER2 STAGGER4; do
  ( cd /tmp && NEURITE_AMD_LIB=$GRAFT_REPO_ROOT/tools/lab/libnrt_fused_$v.so timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s9_fetch_$v -o f -- python $GRAFT_REPO_ROOT/tools/fused_small.py 4 > /dev/null 2>&1 )
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
tot = n = 0
for f in glob.glob('gpurun_out/s9_fetch_%s/**/*counter_collection.csv' % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'warp_dice_wc' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            tot += float(r['Counter_Value']); n += 1
print(v, 'FETCH_SIZE KiB per dispatch', round(tot / max(n, 1), 1), 'x2 GB', round(tot / max(n, 1) * 2048 / 1e9, 3))
PY
done
