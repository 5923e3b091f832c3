"""LocallyConnected3D forward against the batch size.  The weights are streamed once per pass: <= 2 entries on the vector kernel,
3 .. 8 entries on the matrix-core kernel (NRT_LC_MFMA=0: the vector kernel with <= 4 entries per pass, as before round 4).
`frac` prices ONE stream of the weights (the algorithmic bytes of the layer call), whatever the number of passes."""
import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
dev = torch.device('cuda:0')


def timeit(fn, n=40):
    for _ in range(20): fn()                     # ~5 ms of work before the clock: short runs otherwise time a GPU still at idle clocks
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


CASES = ((48, 16, 16, torch.bfloat16), (48, 16, 32, torch.bfloat16), (40, 16, 16, torch.float32))
if os.environ.get('LC_CASES'):          # e.g. LC_CASES=0,2  LC_BATCHES=4,8
    CASES = tuple(CASES[int(i)] for i in os.environ['LC_CASES'].split(','))
BATCHES = tuple(int(b) for b in os.environ.get('LC_BATCHES', '1,2,3,4,6,8,16').split(','))
for S, cin, filters, dtype in CASES:
    for B in BATCHES:
        torch.manual_seed(1)
        x = torch.randn(B, S, S, S, cin, device=dev).to(dtype)
        layer = ne.layers.LocallyConnected3D(filters, (3, 3, 3), activation='elu').to(dev)
        with torch.no_grad():
            layer(x[:1].float() if dtype != torch.float32 else x[:1])
            if dtype != torch.float32:
                layer.to(dtype)
            ms = timeit(lambda: layer(x))
        wbytes = layer.kernel.numel() * layer.kernel.element_size()
        print(json.dumps({'op': 'LocallyConnected3D %d^3 x %d -> %d filters, %s' % (S, cin, filters, str(dtype).split('.')[-1]), 'batch': B,
                          'weights_GB': round(wbytes / 1e9, 3), 'fwd_ms': round(ms, 3),
                          'weight_stream_frac_of_8TBs': round(wbytes / ms / 1e6 / 8000, 3), 'mfma': os.environ.get('NRT_LC_MFMA', '1'), 'blocks': os.environ.get('NRT_LC_BLOCKS', 'auto')}), flush=True)
