"""LocallyConnected3D forward against the batch size (the weights are streamed once per pass of <= 4 batch entries)."""
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


for S, cin, filters, dtype in ((48, 16, 16, torch.bfloat16), (48, 16, 32, torch.bfloat16), (40, 16, 16, torch.float32)):
    for B in (1, 2, 3, 4, 8):
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
                          'weight_stream_frac_of_8TBs': round(wbytes * ((B + 3) // 4) / ms / 1e6 / 8000, 3)}), flush=True)
