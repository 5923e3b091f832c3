#!/bin/bash
# bench the wave-window gather (fused and drop-in) for a list of tune words:  tools/wdd_sweep.sh "0 4 ..." ["unfused tunes"]
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in $1; do
  timeout 200 python bench.py --no-cpu-baseline --no-unet --steps 20 --tune $(( (1<<29) | t )) 2>gpurun_out/b_$t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused tune $t', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['mean_dice'])"
done
for t in $2; do
  timeout 200 python bench.py --no-cpu-baseline --no-unet --steps 20 --unfused --variant 7 --tune $t 2>gpurun_out/u_$t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused tune $t', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
