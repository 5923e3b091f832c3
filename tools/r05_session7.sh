#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/s7_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s7_tests.log
timeout 600 python bench.py --unet 2>&1 | grep -o '"fwd_ms": [0-9.]*, "fwd_ms_min": [0-9.]*' | head -4
