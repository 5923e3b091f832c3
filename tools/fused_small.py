"""run the fused warp+Dice kernel on the bench volumes a few times (profiling target):  python tools/fused_small.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
from neurite_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
S, L = 160, 32
mov = torch.stack([synth.one_hot_volume(1 + b, S, L, dev) for b in range(B)])
fix = torch.stack([synth.one_hot_volume(101 + b, S, L, dev) for b in range(B)])
trf = torch.stack([synth.smooth_displacement(7 + b, S, device=dev) for b in range(B)])
for _ in range(6):
    ne.fused.warp_dice(mov, trf, fix)
torch.cuda.synchronize()
