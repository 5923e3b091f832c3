# the driver's N-GPU command with every rank on ONE GPU and the collective over gloo: control flow only (timings mean nothing)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ranks
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NRT_DEVICE=0 NRT_DIST_BACKEND=gloo
for N in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 5 --warmup 2 > gpurun_out/ranks/n$N.json 2> gpurun_out/ranks/n$N.err
  echo "N=$N rc=$?"; python -c "
import json; d=json.loads([l for l in open('gpurun_out/ranks/n$N.json') if l.strip().startswith('{')][-1]); print(d['n_gpus'], d['rccl_ranks'], d['scaling'], d['config']['global_batch'], d['config']['volumes_per_gpu'], d['value'], d['config']['mean_dice'], 'weak' in d and d['weak_per_gpu']['volumes_per_gpu'], d.get('unet_fwd'))" || tail -5 gpurun_out/ranks/n$N.err
done
