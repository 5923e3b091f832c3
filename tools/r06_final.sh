#!/bin/bash
# Round-6 evidence session (one MI355X box, one session):
#   gpurun --timeout 2400 -- 'bash tools/r06_final.sh'
# smoke, all GPU tests, the default bench line (steps pipelined over 3 streams) and the same with --streams 1, rocprofv3 kernel stats of
# both commands + the dispatch trace of the pipelined one (tools/trace_service_time.py), FETCH_SIZE / WRITE_SIZE passes with the ids of the
# library they belong to, L2 / TA / SQ counters of the shipped gather, kernel stats of the unet forward, the stream probes.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
OUT="$ROOT/gpurun_out/r06_final"
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
{ rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12; rocm-smi --showmeminfo vram | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)"; } > "$OUT/info.log" 2>&1
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/tests.log" > "$OUT/tests_gpu_tail.txt"; cat "$OUT/tests_gpu_tail.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-unet > "$OUT/bench_streams1.json" 2>> "$OUT/bench.log"; echo "bench streams1 rc=$?"
timeout 600 python bench.py --rough --no-cpu-baseline --no-unet > "$OUT/bench_rough.json" 2>> "$OUT/bench.log"
FLAGS="--steps 20 --warmup 5 --no-cpu-baseline --no-batch1 --no-unet --no-strong"
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_s1" -o bench -- python "$ROOT/bench.py" $FLAGS --streams 1 > "$OUT/prof_bench_streams1.json" 2> "$OUT/prof.log" < /dev/null ); echo "prof streams1 rc=$?"
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_s3" -o bench -- python "$ROOT/bench.py" $FLAGS > "$OUT/prof_bench_pipelined.json" 2>> "$OUT/prof.log" < /dev/null ); echo "prof pipelined rc=$?"
find "$OUT/prof_s1" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/bench_streams1_kernel_stats.csv"; head -5 "$f"; done
find "$OUT/prof_s3" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/bench_pipelined_kernel_stats.csv"; done
python tools/trace_service_time.py "$OUT/prof_s3" "warp_dice_wc<1, false, false, false, true, true>" > "$OUT/pipelined_trace_summary.json" 2>&1; cat "$OUT/pipelined_trace_summary.json"
python tools/trace_service_time.py "$OUT/prof_s1" "warp_dice_wc<1, false, false, false, true, true>" > "$OUT/streams1_trace_summary.json" 2>&1
# FETCH_SIZE / WRITE_SIZE: serial launches (per-dispatch counters of overlapping kernels would mix)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_$c" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 2 --prewarm-ms 0 --streams 1 --no-cpu-baseline --no-batch1 --no-unet --no-strong > /dev/null 2> "$OUT/pmc_$c.log" < /dev/null ); echo "pmc $c rc=$?"
done
python tools/summarize_pmc.py "$ROOT/gpurun_out" > "$OUT/pmc_summary.log" 2>&1; cp "$ROOT/gpurun_out/pmc_summary.json" "$OUT/pmc_summary.json" 2>/dev/null
python -c "import json; from neurite_amd import build, _lib; print(json.dumps({'library_build_id': _lib.lib().nrt_build_id().decode(), 'tree_build_id': build.build_id(), 'gather_sources_id': build.gather_sources_id()}))" > "$ROOT/gpurun_out/pmc_build_ids.json"; cp "$ROOT/gpurun_out/pmc_build_ids.json" "$OUT/"
# counters of the shipped gather
PMC_ONLY="sq sq2 lds ta l1 l2" bash tools/pmc_cmd.sh wc_r06_final warp_dice_wc python tools/fused_small.py 4 > "$OUT/pmc_wc.log" 2>&1; cp "$ROOT/gpurun_out/pmc_wc_r06_final.json" "$OUT/pmc_wc.json" 2>/dev/null
# unet forward: kernel stats
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_unet" -o unet -- python "$ROOT/tools/unet_small.py" 20 > /dev/null 2> "$OUT/prof_unet.log" < /dev/null )
find "$OUT/prof_unet" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/unet_kernel_stats.csv"; done
# the probes behind DESIGN 4.1 / 6: steps over 1-4 streams at batch 1 .. 32, batch-1 issue forms, few-channel box form
python tools/two_stream_probe.py 4 1 2 8 32 > "$OUT/two_stream_probe.jsonl" 2>/dev/null
python tools/b1_pipeline_probe.py > "$OUT/b1_pipeline_probe.jsonl" 2>/dev/null
python tools/smallc_bench.py > "$OUT/smallc_bench.jsonl" 2>/dev/null
rm -rf "$OUT/prof_s1" "$OUT/prof_s3" "$OUT/prof_unet" "$ROOT/gpurun_out/pmc_FETCH_SIZE" "$ROOT/gpurun_out/pmc_WRITE_SIZE"
du -sh "$OUT"
