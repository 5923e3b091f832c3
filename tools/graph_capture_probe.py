#!/usr/bin/env python3
"""Which part of the unet training step cannot be captured into a hipGraph?   python tools/graph_capture_probe.py <stage> [size]
stages: fwd_nograd, fwd, loss, bwd, sgd (each includes the ones before it).  Prints one JSON line."""
import contextlib
import json
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne        # noqa: E402

stage = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(sys.stderr):
    net = ne.models.unet(16, (size, size, size, 1), 3, 3, 32, feat_mult=2).to(dev)
net.train()
x = torch.randn(1, size, size, size, 1, device=dev)
t = torch.nn.functional.one_hot(torch.randint(0, 32, (1, size, size, size), device=dev), 32).float()
cce, dice = ne.losses.CategoricalCrossentropy(), ne.losses.Dice(check_input_limits=False)
seg = ne.losses.multiple_losses_decorator([cce.loss, dice.loss])
params = list(net.parameters())


def body():
    if stage == 'fwd_nograd':
        with torch.no_grad():
            return net(x)
    y = net(x)
    if stage == 'fwd':
        return y
    loss = seg(t, y).mean()
    if stage == 'loss':
        return loss
    loss.backward()
    if stage == 'bwd':
        return loss
    with torch.no_grad():
        torch._foreach_add_(params, [p.grad for p in params], alpha=-1e-4)
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        for p in params:
            p.grad = None
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for p in params:
    p.grad = None
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = body()
    g.replay()
    torch.cuda.synchronize()
    print(json.dumps({'stage': stage, 'captured': True, 'out_mean': float(out.float().mean())}))
except Exception as e:      # noqa
    tb = traceback.extract_tb(sys.exc_info()[2])
    where = [(os.path.basename(f.filename), f.lineno, f.name) for f in tb if 'neurite_amd' in f.filename or 'tools' in f.filename][-4:]
    print(json.dumps({'stage': stage, 'captured': False, 'error': str(e)[:120], 'where': where}))
