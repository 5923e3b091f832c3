"""Resize(2) of 4 x 80^3 x C fields: the script rocprofv3 runs to separate kernel time from host time"""
import sys, time, torch
sys.path.insert(0, '.')
import neurite_amd as ne
dev = torch.device('cuda:0')
for C in (1, 2, 3, 4):
    rs = ne.layers.Resize(2)
    half = torch.randn(4, 80, 80, 80, C, device=dev)
    for _ in range(3): rs(half)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): rs(half)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('C=%d host issue %.1f us/call, wall %.1f us/call' % (C, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6), flush=True)
