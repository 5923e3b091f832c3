import os, sys, json, time
import torch
sys.path.insert(0, os.getcwd())
import neurite_amd as ne
from neurite_amd import models as nm
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = ne.models.unet(16, (160, 160, 160, 1), 3, 3, 32, feat_mult=2).to(dev)
x = torch.randn(1, 160, 160, 160, 1, device=dev)
orig = nm._Conv.pool_foldable
def timeit(n=30):
    for _ in range(10): net(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(n): net(x)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
out = {}
for rep in range(3):
    nm._Conv.pool_foldable = orig
    y1 = net(x); out['fold_all_%d' % rep] = round(timeit(), 4)
    nm._Conv.pool_foldable = lambda self, xx, v: self.cin == 1 and orig(self, xx, v)
    y2 = net(x); out['fold_first_only_%d' % rep] = round(timeit(), 4)
out['same_bits'] = bool(torch.equal(y1, y2))
print(json.dumps(out))
