#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_dice_cce.py tests/test_gpu_deferred.py tests/test_gpu_backward.py tests/test_gpu_interpn.py tests/test_gpu_distributed.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/s11_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s11_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/s11_bench.json 2> gpurun_out/s11_bench.log; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open('gpurun_out/s11_bench.json').read().strip().splitlines()[-1])
r = j['roofline']
print('value', j['value'], 'ms', j['ms_per_step'], 'frac', r['frac'], 'launch', r['avg_launch_ms'], 'b1', r.get('batch1', {}).get('frac'), r.get('batch1', {}).get('avg_launch_ms'), 'standalone', r['standalone_interpn']['frac'], 'standalone b1', r.get('batch1', {}).get('standalone_interpn_frac'))
print('registration', j['training']['registration_step'], 'strong', j['cfg4_strong']['value'], j['cfg4_strong']['kernel_ms'], 'unet', j['unet_fwd']['fwd_ms'])
PY
