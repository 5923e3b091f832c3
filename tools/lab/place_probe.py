#!/usr/bin/env python3
"""Is the time of the stand-alone warp (4 x 160^3 x 32) a function of WHERE its buffers lie?  Serial launches, HIP events around 20; the output is
a caller-provided region of one big arena at different byte offsets from the moving volume's address (the kernel sees only pointers)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import neurite_amd as ne
from neurite_amd import synth, utils as U, _lib as L
dev = torch.device('cuda:0')
mov, fix, trf = synth.cfg2_batch(4, 160, 32, device=dev)
del fix
n = mov.numel()
print(json.dumps({'mov': hex(mov.data_ptr()), 'trf': hex(trf.data_ptr()), 'mov_mod_2MB': mov.data_ptr() % (2 << 20), 'trf_mod_2MB': trf.data_ptr() % (2 << 20)}), flush=True)
st = ne.layers.SpatialTransformer(interp_method='linear')
ne.deferred.enabled = False


def timeit(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 4)


# the product call (torch allocates the output)
keep = []
def product():
    keep[:] = [st([mov, trf])]
print(json.dumps({'product_call_ms': [timeit(product), timeit(product)], 'out_ptr': hex(keep[0].data_ptr()), 'out_minus_mov_mod_1MB': (keep[0].data_ptr() - mov.data_ptr()) % (1 << 20)}), flush=True)
keep.clear()
# the same kernel writing into an arena at chosen offsets
arena = torch.empty(n + (64 << 20), dtype=torch.float32, device=dev)
base = arena.data_ptr()
import ctypes
real_empty = torch.empty
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 8192, 16 << 20, (16 << 20) + 65536, 37 << 20, 63 << 20):
    out = arena[off // 4: off // 4 + n].view(mov.shape)

    def fake_empty(*shape, **kw):
        sh = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        return out if sh == tuple(mov.shape) else real_empty(*shape, **kw)
    row = {'offset_from_arena_base': off, 'out_minus_mov_mod_4KB': (out.data_ptr() - mov.data_ptr()) % 4096, 'out_minus_mov_mod_2MB': (out.data_ptr() - mov.data_ptr()) % (2 << 20),
           'out_ptr': hex(out.data_ptr())}
    torch.empty = fake_empty
    try:
        w = st([mov, trf])
        row['wrote_into_arena'] = w.data_ptr() == out.data_ptr()
        row['ms'] = [timeit(lambda: st([mov, trf])), timeit(lambda: st([mov, trf]))]
    except Exception as e:      # noqa
        row['error'] = str(e)[:200]
    finally:
        torch.empty = real_empty
    print(json.dumps(row), flush=True)
