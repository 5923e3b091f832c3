"""the fused warp + Dice on a [B, X, Y, Z, 32] problem with a smooth field (std 3 voxels, like the bench's), a few launches: the
profiling target of the over-fetch-against-march-length probes (profiles/r05_lab/fetch_vs_length.json).
    python tools/fetch_vs_length.py X Y Z [B]      e.g. 16 64 256 1: 512 columns = one per resident block, all starting together"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neurite_amd as ne
X, Y, Z = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(5)
mov = torch.rand((B, X, Y, Z, 32), device=dev, generator=g)
fix = torch.rand((B, X, Y, Z, 32), device=dev, generator=g)
coarse = torch.randn((B, 3, max(X // 8, 2), max(Y // 8, 2), max(Z // 8, 2)), device=dev, generator=g)
trf = torch.nn.functional.interpolate(coarse, size=(X, Y, Z), mode='trilinear', align_corners=True)
trf = (trf * (3.0 / float(trf.std()))).permute(0, 2, 3, 4, 1).contiguous()
for _ in range(6):
    d = ne.fused.warp_dice(mov, trf, fix)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ne.fused.warp_dice(mov, trf, fix)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('%d x %d x %d x %d: %.4f ms, %.3f of 8 TB/s on 268 B / voxel' % (B, X, Y, Z, ms, B * X * Y * Z * 268 / ms / 8e9), flush=True)
