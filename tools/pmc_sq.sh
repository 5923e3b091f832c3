#!/bin/bash
# SQ issue counters for one bench configuration:  tools/pmc_sq.sh <name> [bench args]
# (one --pmc pass, kernel-trace only; summary printed and kept in gpurun_out/sq_<name>.json)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
name="$1"; shift
OUT="$ROOT/gpurun_out/sq_$name"
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CTRS="${PMC_CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY}"
( cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -o p -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-unet "$@" > /dev/null 2> "$OUT.log" )
python - "$OUT" "$name" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, name = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '?').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        if not any(s in k for s in ('interpn', 'warp', 'gather')):
            continue
        a = acc[k][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
res = {k: {c: round(s / max(n, 1), 1) for c, (s, n) in v.items()} for k, v in acc.items()}
json.dump(res, open(out + '.json', 'w'), indent=1)
for k, v in res.items():
    print(name, k)
    for c, x in sorted(v.items()):
        print('    %-24s %14.0f' % (c, x))
PY
rm -rf "$OUT"
