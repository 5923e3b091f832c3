"""How fast is a read-only stream whose working set fits the 256 MB Infinity Cache?  (soft Dice kernel, 2 x N bytes)"""
import json, sys, torch
import neurite_amd as ne
dev = torch.device('cuda:0')
D = ne.metrics.Dice(check_input_limits=False)
for mb in (16, 32, 64, 96, 128, 256, 512, 1024):
    V = mb * (1 << 20) // (32 * 4)
    t = torch.rand(1, V, 32, device=dev); p = torch.rand(1, V, 32, device=dev)
    for _ in range(3): D.dice(t, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    n = 20
    e0.record()
    for _ in range(n): D.dice(t, p)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(json.dumps({'MB_each': mb, 'ms': round(ms, 4), 'TBps': round(2 * mb * (1 << 20) / ms / 1e9, 3)}))
