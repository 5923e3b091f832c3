"""Per-phase clock counts of the wave-window gather (gather_wdd.hip, diagnostic bit 3 of its tune word)."""
import sys, torch
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
import os
B = int(os.environ.get('WB', 4)); SZ = int(os.environ.get('WS', 160))
mov, fix, trf = synth.cfg2_batch(B, SZ, 32, device=dev)
for t in [int(a) for a in sys.argv[1:]] or [0]:
    tune = (1 << 29) | (8 << 24) | t
    d, s = ne.fused.warp_dice(mov, trf, fix, return_sums=True, _tune=tune)
    s = s.double().sum(0).cpu()
    ph = [float(s[0, k]) for k in range(4)] + [float(s[1, 0])]
    nsub, nwave = float(s[1, 1]), float(s[1, 2])
    names = ['index+T issue', 'dedup', 'rowlist+DMA issue', 'wait vmcnt', 'blend']
    print('tune', t, 'waves', nwave, 'sub-windows per window %.3f' % (nsub / (B * SZ ** 3 / 64)))
    for n, v in zip(names, ph):
        print('   %-18s %8.0f clk per window' % (n, v / (B * SZ ** 3 / 64)))
    print('   total %.0f' % (sum(ph) / (B * SZ ** 3 / 64)))
