"""Batch 1 (BASELINE config 2 as written) of the fused warp + Dice kernel: x segments (tune bits 16-23) x resident blocks per CU
(NRT_FUSED_LDS_KB: dynamic LDS that caps the blocks per CU) x patch regions.  One process per LDS setting (the library reads
the variable once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    import neurite_amd as ne
    from neurite_amd import synth
    dev = torch.device('cuda:0')
    mov, fix, trf = synth.cfg2_batch(1, 160, 32, device=dev)
    base = 3 | (2 << 4) | (3 << 8) | (1 << 14)
    for lry, lrz in ((3, 2), (2, 2), (3, 3)):
        for nseg in (1, 2, 3, 4, 5, 8, 10, 16):
            t = base | (nseg << 16) | (lry << 24) | (lrz << 27)
            for _ in range(3):
                ne.fused.warp_dice(mov, trf, fix, _tune=t)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20):
                ne.fused.warp_dice(mov, trf, fix, _tune=t)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(json.dumps({'lds_kb': os.environ.get('NRT_FUSED_LDS_KB'), 'region': [lry, lrz], 'nseg': nseg, 'ms': round(ms, 4),
                              'frac': round(160 ** 3 * 268 / ms / 1e6 / 8000, 4)}), flush=True)
else:
    for kb in ('75', '50', '38', '30', '0'):
        env = dict(os.environ, NRT_FUSED_LDS_KB=kb)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, stdin=subprocess.DEVNULL)
