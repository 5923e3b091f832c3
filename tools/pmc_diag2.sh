#!/bin/bash
# second diagnostic set: where the vector-memory pipe (SQ -> TA -> TCP -> TD) stalls
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/diag2"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
GROUPS_=(
 "sqv:SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "ta1:TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"
 "ta2:TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "tcp1:TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"
 "tcp2:TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"
 "td:TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum"
)
for g in "${GROUPS_[@]}"; do
  name="${g%%:*}"; ctrs="${g#*:}"
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/${name}" -o p -- \
      python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-unet "$@" > /dev/null 2> "$OUT/${name}.log" )
  echo "$name rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, '*'))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '?').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
            if not any(s in k for s in ('interpn_z', 'dice_soft_vec', 'warp_dice')):
                continue
            a = acc[k][row['Counter_Name']]
            a[0] += float(row['Counter_Value']); a[1] += 1
    for k, v in acc.items():
        print(os.path.basename(d), k, {c.replace('_sum', ''): '%.3g' % (s / max(n, 1)) for c, (s, n) in v.items()})
PY
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null
