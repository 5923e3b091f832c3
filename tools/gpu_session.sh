#!/bin/bash
# One MI355X session: smoke, GPU parity tests, kernel-variant sweep, headline bench, rocprofv3 summaries.
# Run through:  gpurun --timeout 1800 -- 'bash tools/gpu_session.sh [stages...]'
# Every stage logs to gpurun_out/<stage>.log and never aborts the others.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
STAGES="${*:-info smoke tests sweep bench prof pmc}"
echo "stages: $STAGES" | tee "$OUT/session.log"

stage() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/session.log"; }

for s in $STAGES; do
case $s in
info)
  stage info
  { rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; rocm-smi --showmeminfo vram | head -8; nproc; free -g | head -2; lscpu | grep -E "Model name|^CPU\(s\)"; } > "$OUT/info.log" 2>&1
  ;;
smoke)
  stage smoke
  timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/session.log"
  ;;
tests)
  stage tests
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee -a "$OUT/session.log"
  tail -5 "$OUT/tests.log" | tee -a "$OUT/session.log"
  ;;
tests_all)
  stage tests_all
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee -a "$OUT/session.log"
  tail -15 "$OUT/tests.log" | tee -a "$OUT/session.log"
  ;;
sweep)
  stage sweep
  timeout 600 python bench.py --sweep > "$OUT/sweep_smooth.log" 2>&1; echo "sweep rc=$?" | tee -a "$OUT/session.log"
  timeout 600 python bench.py --sweep --rough > "$OUT/sweep_rough.log" 2>&1
  timeout 600 python bench.py --sweep --batch-per-gpu 1 > "$OUT/sweep_smooth_b1.log" 2>&1
  grep -h '"kernel"' "$OUT/sweep_smooth.log" | tee -a "$OUT/session.log"
  ;;
bench)
  stage bench
  timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?" | tee -a "$OUT/session.log"
  cat "$OUT/bench.json" | tee -a "$OUT/session.log"
  timeout 600 python bench.py --unfused --no-cpu-baseline > "$OUT/bench_unfused.json" 2>> "$OUT/bench.log"
  timeout 600 python bench.py --batch-per-gpu 1 --no-cpu-baseline > "$OUT/bench_b1.json" 2>> "$OUT/bench.log"
  timeout 600 python bench.py --rough --no-cpu-baseline > "$OUT/bench_rough.json" 2>> "$OUT/bench.log"
  cat "$OUT/bench_unfused.json" "$OUT/bench_b1.json" "$OUT/bench_rough.json" | tee -a "$OUT/session.log"
  ;;
prof)
  stage prof
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- \
      python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-unet --no-strong > "$OUT/prof_bench.json" 2> "$OUT/prof.log" < /dev/null )
  echo "prof rc=$?" | tee -a "$OUT/session.log"
  ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_unfused" -o bench -- \
      python "$ROOT/bench.py" --unfused --steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-unet --no-strong > "$OUT/prof_bench_unfused.json" 2>> "$OUT/prof.log" < /dev/null )
  find "$OUT/prof" -name "*kernel_stats.csv" | head -3 | while read f; do echo "$f"; head -12 "$f"; done | tee -a "$OUT/session.log"
  ;;
dist1)
  # the multi-process code path on one GPU: process group of size 1 over RCCL, with and without hipGraph capture of the step
  stage dist1
  for extra in "" "--graph"; do
    NRT_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-unet $extra \
        > "$OUT/bench_dist1${extra}.json" 2> "$OUT/bench_dist1${extra}.log"
    echo "dist1 $extra rc=$?" | tee -a "$OUT/session.log"
    python - "$OUT/bench_dist1${extra}.json" <<'PY' | tee -a "$OUT/session.log"
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: j[k] for k in ('value', 'ms_per_step', 'rccl_ranks', 'ms_per_step_per_rank')}, j['config'].get('step_launch'), j['roofline']['avg_launch_ms'])
except Exception as e:
    print('no json', e)
PY
  done
  timeout 600 python bench.py --graph --no-cpu-baseline --no-unet > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.log"; echo "graph rc=$?" | tee -a "$OUT/session.log"
  ;;
pmcsweep)
  stage pmcsweep
  for c in FETCH_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmcsweep_$c" -o sweep -- \
        python "$ROOT/bench.py" --sweep --batch-per-gpu 1 > /dev/null 2> "$OUT/pmcsweep_$c.log" )
    echo "pmcsweep $c rc=$?" | tee -a "$OUT/session.log"
  done
  python tools/summarize_pmc.py "$OUT" sweep 2>&1 | tail -60 | tee -a "$OUT/session.log"
  ;;
pmc)
  stage pmc
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -k 5 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o bench -- \
        python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-unet --no-strong > /dev/null 2> "$OUT/pmc_$c.log" < /dev/null )
    echo "pmc $c rc=$?" | tee -a "$OUT/session.log"
  done
  python tools/summarize_pmc.py "$OUT" 2>&1 | tee -a "$OUT/session.log"
  # which binary the counters belong to: the id of the library that ran and of the gather's sources (tools/update_hbm_traffic.py)
  python -c "import json; from neurite_amd import build, _lib; print(json.dumps({'library_build_id': _lib.lib().nrt_build_id().decode(), 'tree_build_id': build.build_id(), 'gather_sources_id': build.gather_sources_id()}))" > "$OUT/pmc_build_ids.json" 2>> "$OUT/session.log"
  cat "$OUT/pmc_build_ids.json" | tee -a "$OUT/session.log"
  ;;
esac
done
# keep the merged output small: drop bulky traces, keep csv summaries
find "$OUT" -name "*.db" -size +8M -delete 2>/dev/null
du -sh "$OUT" | tee -a "$OUT/session.log"
