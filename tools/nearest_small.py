"""run the nearest-neighbour warp of the bench volumes a few times (profiling target):  python tools/nearest_small.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
S, L, B = 160, 32, 4
mov = torch.stack([synth.one_hot_volume(1 + b, S, L, dev) for b in range(B)])
trf = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
st = ne.layers.SpatialTransformer(interp_method='nearest')
for _ in range(6):
    ne.deferred.materialize(st([mov, trf]))
torch.cuda.synchronize()
