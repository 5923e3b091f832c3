#!/bin/bash
# Counter passes (one rocprofv3 --pmc group per pass, kernel-trace only) over an arbitrary command:
#   gpurun -- 'bash tools/pmc_cmd.sh <name> <kernel-substring> python tools/warp_small.py 1'
# -> gpurun_out/pmc_<name>.json (mean counter value per dispatch for kernels whose name contains the substring)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
name="$1"; match="$2"; shift 2
OUT="$ROOT/gpurun_out/pmc_$name"
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 PYTHONPATH="$ROOT"
GROUPS_=(
 "l1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "l2:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
 "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "sq2:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES"
 "ta:TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "lds:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS"
 "mfma:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"
 "mfma2:SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"

)
for g in "${GROUPS_[@]}"; do
  gname="${g%%:*}"; ctrs="${g#*:}"
  if [ -n "${PMC_ONLY:-}" ] && [[ " $PMC_ONLY " != *" $gname "* ]]; then continue; fi      # PMC_ONLY="l1 l2": just these groups
  ( cd "$ROOT" && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/$gname" -o p -- "$@" > /dev/null 2> "$OUT/$gname.log" )
  echo "$gname rc=$?"
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
