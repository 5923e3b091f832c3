#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
NEURITE_AMD_LIB=$PWD/tools/lab/libnrt_fused_BC161_OOR1.so timeout 600 python -m pytest tests/test_gpu_dice_cce.py tests/test_gpu_interpn.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "fused or wave_cache or full_size" > gpurun_out/s4_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s4_tests.log
FUSED_VARIANTS="PROBE=0 BC16=1 BC16=1,OOR=1" FUSED_REPS=3 FUSED_STEPS=100 python tools/fused_variants.py > gpurun_out/s4_variants.jsonl 2>&1; cat gpurun_out/s4_variants.jsonl
