set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s5
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s5/bench_$i.json 2> gpurun_out/s5/bench_$i.err; echo "bench $i rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/s5/bench_$i.json')); r=d['roofline']; sa=r['standalone_interpn']; print(d['value'], r['frac'], r['traffic'], 'iso', r['isolated_launch']['frac'], 'b1', r['batch1']['frac'], r['batch1']['isolated_frac'], 'SA', sa['frac'], sa['avg_launch_ms'], sa['pipelined']['frac'], sa['batch32']['frac'], sa['batch32']['ms_per_volume'], 'strong', d['value_strong_b32'], 'unet', d['unet_fwd']['fwd_ms'], 'reg', d['training']['registration_step']['ms'])"; done
