#!/usr/bin/env python3
"""Does a hipGraph replay of the fused warp + Dice launch recompute, or does it leave stale partial sums behind?  The inputs are changed
between replays (a correct replay follows them).   python tools/graph_fused_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne              # noqa: E402
from neurite_amd import synth        # noqa: E402

dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev)
fix_a, fix_b = fix.clone(), torch.roll(fix, 7, dims=-1).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, tune in (('wc', 0), ('reg', 1 << 30)):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            ne.fused.warp_dice(mov, trf, fix, _tune=tune)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        d = ne.fused.warp_dice(mov, trf, fix, _tune=tune)
    res = {}
    for tag, src in (('a', fix_a), ('b', fix_b), ('a2', fix_a)):
        fix.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        got = d.clone()
        want = ne.fused.warp_dice(mov, trf, fix, _tune=tune)
        res[tag] = float((got - want).abs().max())
    fix.copy_(fix_a)
    print(json.dumps({'kernel': name, 'max_abs_diff_replay_vs_eager_after_changing_the_fixed_map': res, 'ms_replay': round(timeit(g.replay), 4),
                      'ms_eager': round(timeit(lambda: ne.fused.warp_dice(mov, trf, fix, _tune=tune)), 4)}))
