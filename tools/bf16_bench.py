"""fused warp + Dice on label maps stored as bfloat16 vs float32 (4 x 160^3 x 32, sigma = 3 field):  python tools/bf16_bench.py"""
import json, sys, torch
sys.path.insert(0, '.')
import neurite_amd as ne
from neurite_amd import synth
dev = torch.device('cuda:0')
B, S, L = 4, 160, 32
mov, fix, trf = synth.cfg2_batch(B, S, L, device=dev, seed0=100)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
V = S ** 3
for name, m, f, bpv in (('float32', mov, fix, 4 * L + 12 + 4 * L), ('bfloat16 storage', mov.bfloat16(), fix.bfloat16(), 2 * L + 12 + 2 * L)):
    ms = timeit(lambda: ne.fused.warp_dice(m, trf, f))
    print(json.dumps({'maps': name, 'ms_per_step': round(ms, 4), 'Mvoxels_per_s': round(B * V / ms / 1e3, 1),
                      'algorithmic_B_per_voxel': bpv, 'frac_of_hbm_peak': round(bpv * B * V / ms / 1e6 / 8000, 4)}))
