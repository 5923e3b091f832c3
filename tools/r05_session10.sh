#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in "$@"; do
  ( cd /tmp && NEURITE_AMD_LIB=$GRAFT_REPO_ROOT/tools/lab/$v timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s10_fetch_$v -o f -- python $GRAFT_REPO_ROOT/tools/fused_small.py 4 > /dev/null 2>&1 )
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
acc = {}
for f in glob.glob('gpurun_out/s10_fetch_%s/**/*counter_collection.csv' % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and ('warp_dice' in r['Kernel_Name']):
            k = r['Kernel_Name'].split('(')[0][-60:]
            a = acc.setdefault(k, [0.0, 0]); a[0] += float(r['Counter_Value']); a[1] += 1
for k, (t, n) in acc.items():
    print(v, k, 'FETCH x2 GB', round(t / n * 2048 / 1e9, 3), 'n', n)
PY
done
