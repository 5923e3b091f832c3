#!/usr/bin/env python3
"""Independent steps of the fused warp + Dice issued round-robin on 1, 2, 3 streams: does the head of step k + 1 fill the tail of step k?
(VERDICT r5 item 2).  ms per step = wall time of N steps / N between synchronizes; every step has its own output, results are compared
with the single-stream run.    python tools/two_stream_probe.py [batch ...]      (GPU box)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
N = 48
for batch in [int(a) for a in sys.argv[1:]] or [4, 1]:
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    want = ne.fused.warp_dice(mov, trf, fix).clone()
    row = {'batch': batch}
    for ns in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        outs = []
        def run(n):
            outs.clear()
            for k in range(n):
                with torch.cuda.stream(streams[k % ns]):
                    outs.append(ne.fused.warp_dice(mov, trf, fix))
        run(8); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(N); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / N * 1e3)
        ok = all(torch.equal(o, want) for o in outs)
        row['ms_per_step_%d_streams' % ns] = round(best, 4)
        row['frac_%d' % ns] = round(batch * 160 ** 3 * 268 / best / 1e6 / 8000.0, 4)
        row['same_bits_%d' % ns] = ok
    print(json.dumps(row), flush=True)
    del mov, fix, trf
