"""Backward passes at channel counts off the tuned paths, 4 x 160^3: which generic kernel is the slow one"""
import json, sys, warnings, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
warnings.simplefilter('ignore')
def timeit(fn, n=2):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for C in (3, 5, 20):
    vol = torch.rand(B, S, S, S, C, device=dev)
    other = torch.rand(B, S, S, S, C, device=dev)
    p = torch.softmax(other, -1)
    st = ne.layers.SpatialTransformer()
    g = torch.randn_like(vol)
    res = {'C': C}
    fg = flow.clone().requires_grad_()
    out = st([vol, fg])
    res['warp_bwd_flow_ms'] = timeit(lambda: torch.autograd.grad(out, fg, g, retain_graph=True))
    vg = vol.clone().requires_grad_()
    out2 = st([vg, flow])
    res['warp_bwd_vol_ms'] = timeit(lambda: torch.autograd.grad(out2, vg, g, retain_graph=True))
    pg = vol.clone().requires_grad_()
    d = ne.metrics.Dice().loss(other, pg).sum()
    res['dice_bwd_ms'] = timeit(lambda: torch.autograd.grad(d, pg, retain_graph=True))
    qg = p.clone().requires_grad_()
    l = ne.metrics.CategoricalCrossentropy()(vol, qg)
    res['cce_fwd_ms'] = timeit(lambda: ne.metrics.CategoricalCrossentropy()(vol, p))
    res['cce_bwd_ms'] = timeit(lambda: torch.autograd.grad(l, qg, retain_graph=True))
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
