"""
Data-parallel training of neurite's unet on MI355X: one process per GPU, RCCL gradient all-reduce over xGMI.

    python examples/train_unet_dp.py --steps 5                                    # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_unet_dp.py

Every rank synthesises its own (image, one-hot segmentation) pairs on the device with the label-to-image model
(`ne.models.labels_to_image`), runs forward + backward of `ne.models.unet` on the HIP kernels (`model.train()`), averages
the gradients over the ranks with ONE all-reduce of the flat buffer the gradients live in (`ne.distributed.GradientBucket`: no packing,
no copy back) and applies SGD.
"""

import argparse
import contextlib
import io
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne  # noqa: E402
from neurite_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--size', type=int, default=96)
    ap.add_argument('--labels', type=int, default=8)
    ap.add_argument('--lr', type=float, default=1e-2)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    S, L = args.size, args.labels
    label_values = list(range(L))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(0)                                         # identical initial weights on every rank
        net = ne.models.unet(8, (S, S, S, 1), 3, 3, L, feat_mult=2).to(dev)
        gen = ne.models.labels_to_image((S, S, S), label_values, warp_std=1.0, seeds={})
    net.train()
    params = list(net.parameters())
    bucket = ne.distributed.GradientBucket(params)                   # every p.grad is now a view into ONE flat float32 buffer
    cce = ne.losses.CategoricalCrossentropy()
    dice = ne.metrics.Dice(check_input_limits=False)
    losses = []
    t0 = time.perf_counter()
    for step in range(args.steps):
        labels = synth.blob_labels(1000 * rank + step, size=S, nb_labels=L, coarse=6, device=dev)[None, ..., None].to(torch.int32)
        image, onehot = gen(labels)                                  # [1, S, S, S, 1], [1, S, S, S, L]
        pred = net(image)
        loss = cce(onehot, pred) - dice.mean_dice(onehot, pred)
        loss.backward()
        bucket.all_reduce()                                          # one RCCL all-reduce of that buffer; no-op at world size 1
        with torch.no_grad():
            for p in params:
                p -= args.lr * p.grad
            bucket.zero_()                                           # (not p.grad = None: the views are the point)
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps({'world': world, 'steps': args.steps, 'size': S, 'labels': L, 'first_loss': round(losses[0], 4),
                          'last_loss': round(losses[-1], 4), 's_per_step': round(dt / args.steps, 4)}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
