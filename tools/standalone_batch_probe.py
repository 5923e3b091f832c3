#!/usr/bin/env python3
"""The stand-alone warp (SpatialTransformer linear, 160^3 x 32 float32, the op that WRITES the warped volume; nrt_interpn_f32 variant 10)
at batch 1 .. 32, launches strictly serial on one stream and round-robin on 3 streams: ms per launch / per volume and the fraction of the
HBM roof on 268 B per voxel.  north_star's 60 % target is 0.229 ms per volume.
    python tools/standalone_batch_probe.py [batch ...]      (GPU box)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
st = ne.layers.SpatialTransformer(interp_method='linear')
keep = ne.deferred.enabled
ne.deferred.enabled = False
V = 160 ** 3
for batch in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32]:
    mov, fix, trf = synth.cfg2_batch(batch, 160, 32, device=dev)
    del fix
    N = max(6, 96 // batch)
    row = {'batch': batch, 'launches': N}
    for ns in (1, 3):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        outs = [None] * ns
        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % ns]):
                    outs[k % ns] = st([mov, trf])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            run(ns * 2); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); run(N); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / N * 1e3)
        row['ms_per_launch_%d_streams' % ns] = round(best, 4)
        row['ms_per_volume_%d_streams' % ns] = round(best / batch, 4)
        row['frac_%d' % ns] = round(batch * V * 268 / best / 1e6 / 8000.0, 4)
        del outs
        torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
    del mov, trf
    torch.cuda.empty_cache()
ne.deferred.enabled = keep
