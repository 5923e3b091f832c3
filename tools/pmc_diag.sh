#!/bin/bash
# Diagnostic counter passes over the bench kernels (one rocprofv3 --pmc group per pass; kernel-trace only).
#   gpurun --timeout 900 -- 'bash tools/pmc_diag.sh [extra bench args]'
# Output: gpurun_out/diag/<group>_<mode>/ + gpurun_out/diag/summary.json (mean counter value per dispatch per kernel)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/diag"
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
GROUPS_=(
 "l1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "tlb:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "l2:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
 "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "ta:TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
)
MODES="${PMC_MODES:-fused unfused}"
for g in "${GROUPS_[@]}"; do
  name="${g%%:*}"; ctrs="${g#*:}"
  for mode in $MODES; do
    extra=""; [ "$mode" = unfused ] && extra="--unfused"
    ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/${name}_$mode" -o p -- \
        python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-unet $extra "$@" > /dev/null 2> "$OUT/${name}_$mode.log" )
    echo "$name $mode rc=$?"
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, '*_*'))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '?').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
            if not any(s in k for s in ('interpn', 'dice', 'warp', 'gather')):
                continue
            a = acc[k][row['Counter_Name']]
            a[0] += float(row['Counter_Value']); a[1] += 1
    res[os.path.basename(d)] = {k: {c: round(s / max(n, 1), 1) for c, (s, n) in v.items()} for k, v in acc.items()}
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.db" -delete 2>/dev/null
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
du -sh "$OUT"
