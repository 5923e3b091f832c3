#!/bin/bash
# FETCH_SIZE (true bytes = 2 x counter on gfx950) of the fused kernel for a list of "LDS_KB:tune" settings
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd /tmp
for cfg in "$@"; do
  kb="${cfg%%:*}"; t="${cfg#*:}"
  rm -rf /tmp/fp_$$; 
  NRT_FUSED_LDS_KB=$kb rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fp_$$ -o s -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unet --tune $t > /dev/null 2> /tmp/fp_$$.log
  python - <<PY
import csv,glob
f=glob.glob('/tmp/fp_$$/**/*counter_collection.csv', recursive=True)[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name']=='FETCH_SIZE' and 'warp_dice' in r['Kernel_Name']]
print('lds_kb=$kb tune=$t FETCH true GB', round(2*sum(v)/len(v)*1024/1e9,3), 'n', len(v))
PY
done
