"""Channel / label counts off the tuned paths (the generic kernels): ms and fraction of the HBM roof at 4 x 160^3 -- a sweep for
pathologies (per-voxel global atomics, scratch arrays, uncoalesced rows)."""
import json, sys, warnings, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
warnings.simplefilter('ignore')
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def row(op, ms, nbytes):
    print(json.dumps({'op': op, 'ms': round(ms, 3), 'frac': round(nbytes / ms / 1e6 / 8000, 3)}), flush=True)
S, B = 160, 4
nvox = B * S ** 3
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for C in (5, 7, 12, 20, 33):
    vol = torch.rand(B, S, S, S, C, device=dev)
    other = torch.rand(B, S, S, S, C, device=dev)
    st = ne.layers.SpatialTransformer()
    row('linear warp C=%d' % C, timeit(lambda: ne.deferred.materialize(st([vol, flow]))), nvox * (8 * C + 12))
    stn = ne.layers.SpatialTransformer(interp_method='nearest')
    row('nearest warp C=%d' % C, timeit(lambda: stn([vol, flow])), nvox * (8 * C + 12))
    row('soft Dice L=%d' % C, timeit(lambda: ne.metrics.Dice().dice(vol, other)), nvox * 8 * C)
    row('hard Dice (prob) L=%d' % C, timeit(lambda: ne.metrics.HardDice(C, input_type='prob').dice(vol, other)), nvox * 8 * C)
    p = torch.softmax(other, -1)
    row('CCE C=%d' % C, timeit(lambda: ne.metrics.CategoricalCrossentropy()(vol, p)), nvox * 8 * C)
    vg = vol.clone().requires_grad_()
    fg = flow.clone().requires_grad_()
    def step():
        vg.grad = None; fg.grad = None
        ne.metrics.Dice().loss(other, st([vg, fg])).sum().backward()
    row('warp + Dice loss fwd+bwd (both grads) C=%d' % C, timeit(step, n=2), nvox * 8 * C)
    del vol, other, p, vg
