"""run the few-channel linear warp a few times (profiling target):  python tools/warp_small.py [C] [variant]"""
import sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
S, B = 160, 4
flow = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
vol = torch.randn(B, S, S, S, C, device=dev)
st = ne.layers.SpatialTransformer()
st._variant = variant
for _ in range(6):
    ne.deferred.materialize(st([vol, flow]))
torch.cuda.synchronize()
