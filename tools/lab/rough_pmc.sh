# HBM-side bytes of the fused gather on the worst-case field U(-80, 80) (bench.py --rough): FETCH_SIZE / WRITE_SIZE passes + L2 counters
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"; OUT="$ROOT/gpurun_out/rough"; mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CMD="python $ROOT/bench.py --rough --steps 5 --warmup 2 --prewarm-ms 0 --streams 1 --no-cpu-baseline --no-batch1 --no-unet --no-strong"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_$c" -o bench -- $CMD > /dev/null 2> "$OUT/pmc_$c.log" < /dev/null ); echo "pmc $c rc=$?"
done
python tools/summarize_pmc.py "$ROOT/gpurun_out" > "$OUT/pmc_summary.log" 2>&1; cp "$ROOT/gpurun_out/pmc_summary.json" "$OUT/pmc_summary_rough.json"
( cd /tmp && timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/l2" -o bench -- $CMD > /dev/null 2> "$OUT/pmc_l2.log" < /dev/null ); echo "l2 rc=$?"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, 'l2', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '?').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        if 'warp_dice' not in k: continue
        a = acc[k][row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
res = {k: {c: round(s / max(n, 1), 1) for c, (s, n) in v.items()} for k, v in acc.items()}
json.dump(res, open(os.path.join(out, 'l2_rough.json'), 'w'), indent=1); print(json.dumps(res))
PY
timeout 300 $CMD > "$OUT/bench_rough.json" 2> /dev/null; python -c "
import json; d=json.load(open('$OUT/bench_rough.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
cat "$OUT/pmc_summary_rough.json"
rm -rf "$OUT/l2" "$ROOT/gpurun_out/pmc_FETCH_SIZE" "$ROOT/gpurun_out/pmc_WRITE_SIZE"
