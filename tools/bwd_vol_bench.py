"""d out / d vol of the linear warp at 160^3 x 32 (one volume), bench field and worst-case field: NRT_BWD_VOL_DEDUP = 0 plain
scatter, 1 LDS row-accumulator table, 2 counting-sort merge; the time includes the zero fill of the gradient (524 MB)"""
import json, os, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rough in (False, True):
    mov, fix, flow = synth.cfg2_batch(1, 160, 32, device=dev, rough=rough)
    mov_g = mov.clone().requires_grad_()
    out = ne.layers.SpatialTransformer()([mov_g, flow])
    g = torch.randn_like(out)
    ms = timeit(lambda: torch.autograd.grad(out, mov_g, g, retain_graph=True))
    print(json.dumps({'mode': os.environ.get('NRT_BWD_VOL_DEDUP', 'default'), 'field': 'rough U(-80,80)' if rough else 'bench',
                      'grad_vol_ms_incl_zero_fill': round(ms, 3)}))
