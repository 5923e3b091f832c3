"""write-only / read-only / copy rates of plain streams on this box (torch kernels and our own float4 copy kernel)"""
import json, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
dev = torch.device('cuda:0')
n = 512 * 1024 * 1024            # 2 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def timeit(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
gb = n * 4 / 1e9
ms = timeit(lambda: x.fill_(1.0)); print(json.dumps({'op': 'torch fill_ (write only)', 'ms': round(ms, 3), 'TBs': round(gb / ms, 3)}))
ms = timeit(lambda: x.zero_()); print(json.dumps({'op': 'torch zero_ (write only)', 'ms': round(ms, 3), 'TBs': round(gb / ms, 3)}))
ms = timeit(lambda: x.sum()); print(json.dumps({'op': 'torch sum (read only)', 'ms': round(ms, 3), 'TBs': round(gb / ms, 3)}))
ms = timeit(lambda: y.copy_(x)); print(json.dumps({'op': 'torch copy_ (read + write)', 'ms': round(ms, 3), 'TBs_total': round(2 * gb / ms, 3)}))
lib = ne._lib.lib()
for nt in (0, 1):
    for blocks in (2048, 8192):
        ms = timeit(lambda: lib.nrt_membench_copy_f32(ne._lib.ptr(x), ne._lib.ptr(y), n, nt, blocks, ne._lib.stream_ptr(dev)))
        print(json.dumps({'op': 'nrt_membench_copy nt=%d blocks=%d' % (nt, blocks), 'ms': round(ms, 3), 'TBs_total': round(2 * gb / ms, 3)}))
