cd /tmp
for t in 1075 819 1058 530 $(( (1<<13)|(40<<16) )) 4147 1092 ; do
  NRT_SWEEP_COHERENT=0 NRT_SWEEP=3:20 NRT_FUSED_SWEEP=$t rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fs_$t -o s -- python $GRAFT_REPO_ROOT/bench.py --sweep > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/fs_$t.log
  python - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/fs_$t/*counter_collection.csv')[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name']=='FETCH_SIZE' and 'warp_dice' in r['Kernel_Name']]
import re
ms=[l for l in open('$GRAFT_REPO_ROOT/gpurun_out/fs_$t.log') if 'fused_warp_dice' in l]
print($t, 'FETCH true GB', round(2*sum(v)/len(v)*1024/1e9,3), ms[0].strip()[:160] if ms else '')
PY
done
