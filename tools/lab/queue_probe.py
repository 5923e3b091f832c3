#!/usr/bin/env python3
"""Does work on the default stream run slower once other streams of the process have been used?  The stand-alone warp (4 x 160^3 x 32), serial
launches on the default stream, HIP events around 20 launches: (a) in a fresh process, (b) after three side streams have each run a few
launches, (c) the same launches issued on ONE of the side streams, (d) default stream again."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
st = ne.layers.SpatialTransformer(interp_method='linear')
ne.deferred.enabled = False
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev)


def serial(n=20, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(4):
                w = st([mov, trf])
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            w = st([mov, trf])
        e1.record()
        torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


out = {'a_fresh_default_stream': [serial(), serial()]}
side = [torch.cuda.Stream() for _ in range(3)]
for k in range(12):
    with torch.cuda.stream(side[k % 3]):
        w = st([mov, trf])
torch.cuda.synchronize()
out['b_default_stream_after_side_streams_ran'] = [serial(), serial()]
out['c_on_a_side_stream'] = [serial(stream=side[0]), serial(stream=side[1])]
out['d_default_stream_again'] = [serial(), serial()]
del side
torch.cuda.synchronize()
out['e_default_stream_side_streams_dropped'] = [serial()]
print(json.dumps(out))
