"""interpn on volumes that are not float32 (csrc/interpn_any.hip, the coverage path): 4 x 160^3 x C warps"""
import json, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for C in (1, 32):
    for dt in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
        vol = torch.randn(B, S, S, S, C, device=dev).to(dt)
        st = ne.layers.SpatialTransformer()
        ms = timeit(lambda: ne.deferred.materialize(st([vol, flow.to(dt) if dt == torch.float64 else flow])))
        print(json.dumps({'C': C, 'dtype': str(dt), 'ms': round(ms, 3)}), flush=True)
