"""One few-channel warp / resize / VecInt workload for a counter pass (tools/pmc_cmd.sh):  python tools/smallc_pmc_probe.py warp1|warp3|resize1|resize3|vecint"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'warp1'
S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
if what.startswith('warp'):
    C = int(what[4:])
    vol = torch.randn(B, S, S, S, C, device=dev)
    st = ne.layers.SpatialTransformer()
    f = lambda: ne.deferred.materialize(st([vol, flow]))
elif what.startswith('resize'):
    C = int(what[6:])
    half = torch.randn(B, 80, 80, 80, C, device=dev)
    rs = ne.layers.Resize(2)
    f = lambda: rs(half)
else:
    vi = ne.layers.VecInt(int_steps=7)
    f = lambda: vi(flow)
for _ in range(6):
    f()
torch.cuda.synchronize()
