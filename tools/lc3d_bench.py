"""LocallyConnected3D at BASELINE config 5: forward (bench.lc3d_bench) and backward (grad wrt weights / + input)."""
import json, sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
import neurite_amd as ne

dev = torch.device('cuda:0')
res = bench.lc3d_bench(dev)
if len(sys.argv) > 1 and sys.argv[1] == 'bwd':
    torch.manual_seed(6)
    x = torch.randn(1, 96, 96, 96, 16, device=dev, dtype=torch.bfloat16)
    layer = ne.layers.LocallyConnected3D(16, (3, 3, 3))
    y = layer(x)
    g = torch.randn_like(y)

    def run(with_x):
        xg = x.clone().requires_grad_(with_x)
        out = layer(xg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        out.backward(g)
        e1.record(); torch.cuda.synchronize()
        layer.kernel.grad = None; layer.bias.grad = None
        return e0.elapsed_time(e1)
    run(False); run(True)
    res['bwd_weights_ms'] = round(min(run(False) for _ in range(3)), 3)
    res['bwd_weights_and_input_ms'] = round(min(run(True) for _ in range(3)), 3)
    wb = layer.kernel.numel() * 2
    res['bwd_weights_GBs_written'] = round(wb / res['bwd_weights_ms'] / 1e6, 1)
print(json.dumps(res))
