set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s2/bench_$i.json 2> gpurun_out/s2/bench_$i.err; echo "bench $i rc=$?"; tail -3 gpurun_out/s2/bench_$i.err; python -c "
import json; d=json.load(open('gpurun_out/s2/bench_$i.json')); r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r['isolated_launch']['frac'], r['batch1']['frac']); print(json.dumps(r['standalone_interpn']))"; done
