#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
NEURITE_AMD_LIB=$PWD/tools/lab/libnrt_fused_${1:-IFIRST3}.so timeout 600 python -m pytest tests/test_gpu_dice_cce.py tests/test_gpu_interpn.py tests/test_gpu_deferred.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "fused or wave_cache or full_size or deferred" > gpurun_out/s5_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s5_tests.log
FUSED_VARIANTS="${2:-IFIRST=2 IFIRST=3}" FUSED_REPS=3 FUSED_STEPS=60 python tools/fused_variants.py 2>&1 | tee gpurun_out/s5_variants.jsonl
