"""run the BASELINE config 3 unet forward a few times (profiling target):  python tools/unet_small.py [reps]"""
import contextlib
import io
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter('ignore')
    torch.manual_seed(0)
    net = ne.models.unet(16, (160, 160, 160, 1), 3, 3, 32, feat_mult=2).to(dev)
x = torch.randn(1, 160, 160, 160, 1, device=dev)
for _ in range(reps):
    y = net(x)
torch.cuda.synchronize()
