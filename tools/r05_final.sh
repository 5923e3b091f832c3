#!/bin/bash
# Round-5 evidence session (one MI355X box, one session): smoke, all GPU tests, the bench line, rocprofv3 kernel stats of the same
# command, FETCH_SIZE / WRITE_SIZE passes, SQ / LDS / TA counters of the shipped gather, kernel stats + MFMA counters of the unet forward.
#   gpurun --timeout 2400 -- 'bash tools/r05_final.sh'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
OUT="$ROOT/gpurun_out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
bash tools/gpu_session.sh info smoke tests_all bench prof pmc > /dev/null 2>&1
grep -v "^{" "$OUT/session.log" | grep -v "^$" | tail -30
# counters of the shipped gather kernels (fused: batch 4 and 1; stand-alone warp)
PMC_ONLY="sq sq2 lds ta l1 l2" bash tools/pmc_cmd.sh wc_r05_final warp_dice_wc python tools/fused_small.py 4 > "$OUT/pmc_wc_r05_final.log" 2>&1
# unet forward: kernel stats and matrix-core counters
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_unet" -o unet -- python "$ROOT/tools/unet_small.py" 20 > /dev/null 2> "$OUT/prof_unet.log" < /dev/null )
find "$OUT/prof_unet" -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" "$OUT/unet_kernel_stats.csv"; head -14 "$f"; done
PMC_ONLY="mfma mfma2" bash tools/pmc_cmd.sh unet_r05 conv python tools/unet_small.py 5 > "$OUT/pmc_unet_r05.log" 2>&1
tail -5 "$OUT/pmc_unet_r05.log"
find "$OUT" -name "*.db" -size +8M -delete 2>/dev/null
du -sh "$OUT"
