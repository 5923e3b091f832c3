"""d loss / d field of the fused warp + Dice at the bench shape: the wave-cache gather against the register-pipelined kernel
(NRT_BWD_WC=0), events around 10 backward calls each.  BB = batch."""
import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import neurite_amd as ne
dev = torch.device('cuda:0')
B, S, L = int(os.environ.get('BB', 4)), (160, 160, 160), 32
from neurite_amd import synth
mov, fix, flow = synth.cfg2_batch(B, S[0], L, device=dev, seed0=100)     # bench.py's maps and field
flow = flow.clone().requires_grad_(True)
for wc in os.environ.get('WCS', '1 0 1 0').split():
    os.environ['NRT_BWD_WC'] = wc
    d = ne.fused.warp_dice(mov, flow, fix)
    loss = -d.mean()
    for _ in range(3):
        flow.grad = None
        loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        flow.grad = None
        loss.backward(retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print('NRT_BWD_WC=%s B=%d fused backward %.3f ms' % (wc, B, e0.elapsed_time(e1) / 10), flush=True)
