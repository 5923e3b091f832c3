import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neurite_amd import models as nm
dev = torch.device('cuda:0')
torch.manual_seed(0)
cases = [(16, 16, (16, 16, 160), 2), (16, 16, (16, 16, 320), 2), (32, 32, (16, 16, 160), 2), (16, 32, (80, 80, 80), 1)]
if len(sys.argv) > 1:
    cases = [cases[int(sys.argv[1])]]
for (cin, cout, S, B) in cases:
    conv = nm._Conv('c', cin, cout, (3, 3, 3), 1, 'same', 'elu').to(dev)
    x = torch.randn(B, *S, cin, device=dev)
    with torch.no_grad():
        y2 = conv(x, variant=2)
        torch.cuda.synchronize()
        print('case', cin, cout, S, B, 'launching variant 5', flush=True)
        y5 = conv(x, variant=5)
        torch.cuda.synchronize()
        print('   max diff', float((y5 - y2).abs().max()), flush=True)
