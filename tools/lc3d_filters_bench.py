"""LocallyConnected3D beyond BASELINE config 5: 32 filters (27 row groups per lane in bfloat16 -- the two-waves-per-position form
of csrc/lc3d.hip) and float32, forward and backward, ms and fraction of the HBM roof of the weight stream."""
import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
dev = torch.device('cuda:0')


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for S, cin, filters, dtype, B in ((64, 16, 32, torch.bfloat16, 1), (64, 16, 32, torch.bfloat16, 4), (64, 16, 16, torch.bfloat16, 1),
                                  (48, 16, 32, torch.float32, 1), (64, 8, 32, torch.bfloat16, 2)):
    torch.manual_seed(1)
    x = torch.randn(B, S, S, S, cin, device=dev).to(dtype)
    layer = ne.layers.LocallyConnected3D(filters, (3, 3, 3), activation='elu').to(dev)
    with torch.no_grad():
        y = layer(x)
        if dtype != torch.float32:
            layer.to(dtype)
    wbytes = layer.kernel.numel() * layer.kernel.element_size()
    with torch.no_grad():
        ms = timeit(lambda: layer(x))
    g = torch.randn_like(layer(x))

    def bwd():
        xg = x.clone().requires_grad_()
        layer(xg).backward(g)
        layer.kernel.grad = None; layer.bias.grad = None
    msb = timeit(bwd, n=3)
    print(json.dumps({'op': 'LocallyConnected3D %d^3 x %d -> %d filters, %s, batch %d' % (S, cin, filters, str(dtype).split('.')[-1], B),
                      'weights_GB': round(wbytes / 1e9, 3), 'fwd_ms': round(ms, 3), 'fwd_frac_of_8TBs': round(wbytes / ms / 1e6 / 8000, 3),
                      'fwd_plus_bwd_ms': round(msb, 3)}), flush=True)
