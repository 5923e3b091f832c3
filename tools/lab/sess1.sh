set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > gpurun_out/s1/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s1/tests.log
timeout 600 python tools/standalone_batch_probe.py > gpurun_out/s1/standalone_batch.jsonl 2> gpurun_out/s1/standalone_batch.err; echo "probe rc=$?"; cat gpurun_out/s1/standalone_batch.jsonl
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unet > gpurun_out/s1/bench_$i.json 2> gpurun_out/s1/bench_$i.err; echo "bench $i rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/s1/bench_$i.json')); r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r['isolated_launch']['frac'], r['batch1']['frac'], r['standalone_interpn']['frac'], r['standalone_interpn'].get('pipelined',{}).get('frac'))"; done
