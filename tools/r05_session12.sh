#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_dice_cce.py tests/test_gpu_segloss.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider --timeout 600 -k "cce or seg or loss" > gpurun_out/s12_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s12_tests.log
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
for _ in range(2):
    r = bench.lc3d_bench(dev)
    print(json.dumps({k: r[k] for k in ('lc3d_ms', 'frac_of_hbm_peak', 'wcce_ms', 'wcce_ms_as_graph_replay', 'loss')}))
PY
