#!/usr/bin/env python3
"""Batch-1 steps (one 160^3 x 32 volume, ~0.22 ms of device time) issued over 1-3 streams: direct calls against one hipGraph replay per
step, with and without the [sum, count] pair of the step's mean.  Where does a small step's time go?     (GPU box)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth, distributed as nd
dev = torch.device('cuda:0')
N = 96
mov, fix, trf = synth.cfg2_batch(1, 160, 32, device=dev)
want = ne.fused.warp_dice(mov, trf, fix).clone()


def timeit(issue, ns):
    issue(8 * ns); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); issue(N); t1 = time.perf_counter(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e3)
        host = (t1 - t0) / N * 1e3
    return round(best, 4), round(host, 4)


for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    row = {'streams': ns}
    for name, fn in (('direct_dice', lambda: ne.fused.warp_dice(mov, trf, fix)),
                     ('direct_dice_mean', lambda: nd.all_reduce_mean_dice(ne.fused.warp_dice(mov, trf, fix), async_op=True))):
        def issue(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % ns]):
                    fn()
        row[name + '_ms'], row[name + '_host_ms'] = timeit(issue, ns)
    for name, fn in (('graph_dice', lambda: ne.fused.warp_dice(mov, trf, fix)),
                     ('graph_dice_pair', lambda: nd.mean_dice_pair(ne.fused.warp_dice(mov, trf, fix)))):
        graphs = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn(); fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = fn()
            graphs.append((g, out))
        def issue(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % ns]):
                    graphs[k % ns][0].replay()
        row[name + '_ms'], row[name + '_host_ms'] = timeit(issue, ns)
        if name == 'graph_dice':
            torch.cuda.synchronize()
            row['graph_same_bits'] = all(torch.equal(o, want) for _, o in graphs)
    row['frac_best'] = round(160 ** 3 * 268 / min(v for k, v in row.items() if k.endswith('_ms') and 'host' not in k) / 1e6 / 8000.0, 4)
    print(json.dumps(row), flush=True)
