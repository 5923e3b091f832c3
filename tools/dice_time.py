"""time soft Dice on the bench volumes (4 x 160^3 x 32 float32):  python tools/dice_time.py"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
a = torch.stack([synth.one_hot_volume(1 + b, 160, 32, dev) for b in range(4)])
b = torch.stack([synth.one_hot_volume(101 + k, 160, 32, dev) for k in range(4)])
d = ne.metrics.Dice(check_input_limits=False)
for _ in range(3): d.dice(a, b)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): r = d.dice(a, b)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ms = sorted(ts)[2]
print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('NRT_')}, 'ms': round(ms, 4), 'TBs': round(4 * 160 ** 3 * 256 / ms / 1e9, 3),
                  'dice0': float(r[0, 0])}))
