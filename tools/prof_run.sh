#!/bin/bash
# rocprofv3 kernel-trace summary of a python command, safely (never reads stdin, inner timeout, csv output).
#   bash tools/prof_run.sh <outdir-under-gpurun_out> <python args...>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ -f "$ROOT/$1" ]; then set -- "$ROOT/$1" "${@:2}"; fi      # the command runs from /tmp
( cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o t -- python "$@" > "$OUT/run.log" 2>&1 < /dev/null )
echo "rocprof rc=$?"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; head -30 "$f" | cut -c1-200; else echo "no kernel_stats.csv"; tail -5 "$OUT/run.log"; fi
find "$OUT/prof" -name "*.db" -delete 2>/dev/null; find "$OUT/prof" -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
true
