"""d out / d vol of the linear warp at 160^3 x 32 (one volume): LDS row-accumulator kernel vs the plain scatter (NRT_BWD_VOL_DEDUP=0)"""
import json, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
mov, fix, flow = synth.cfg2_batch(1, 160, 32, device=dev)
mov_g = mov.clone().requires_grad_()
out = ne.layers.SpatialTransformer()([mov_g, flow])
g = torch.randn_like(out)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = timeit(lambda: torch.autograd.grad(out, mov_g, g, retain_graph=True))
import os
print(json.dumps({'dedup': os.environ.get('NRT_BWD_VOL_DEDUP', '1'), 'grad_vol_ms_incl_zero_fill': round(ms, 3)}))
