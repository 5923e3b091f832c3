#!/bin/bash
# round-5 session 1: new wave-cache schedule -- parity, then A/B against the round-4 library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_dice_cce.py tests/test_gpu_interpn.py tests/test_gpu_deferred.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/s1_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/s1_tests.log
for rep in 1; do
  timeout 300 python tools/wc_bench.py --batch 4,1 --fields bench,zero,rough > gpurun_out/s1_wc_new_$rep.jsonl 2> gpurun_out/s1_wc_new_$rep.err; echo "new rc=$?"
  NEURITE_AMD_LIB=$PWD/tools/lab/libnrt_r04.so timeout 300 python tools/wc_bench.py --batch 4,1 --fields bench,zero,rough > gpurun_out/s1_wc_old_$rep.jsonl 2> gpurun_out/s1_wc_old_$rep.err; echo "old rc=$?"
done
echo NEW; cat gpurun_out/s1_wc_new_*.jsonl; echo OLD; cat gpurun_out/s1_wc_old_*.jsonl
