#!/bin/bash
# HBM-side bytes of one kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (together they hung a pass
# until its timeout in round 3), kernel-trace only, inner timeouts.
#   bash tools/pmc_mem.sh <name> <kernel-substring> <python script + args>
# -> gpurun_out/pmc_mem_<name>.json  (mean KiB per dispatch as the counters report them; tools/update_hbm_traffic.py applies the
#    gfx950 correction of the microarchitecture guide)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
name="$1"; match="$2"; shift 2
OUT="$ROOT/gpurun_out/pmc_mem_$name"
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
if [ -f "$ROOT/$1" ]; then set -- "$ROOT/$1" "${@:2}"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o p -- python "$@" > /dev/null 2> "$OUT/$c.log" < /dev/null )
  echo "$c rc=$?"
done
python - "$OUT" "$match" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, match = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '?').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        if match not in k:
            continue
        a = acc[k][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
res = {k: {c: round(s / max(n, 1), 1) for c, (s, n) in v.items()} for k, v in acc.items()}
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"
