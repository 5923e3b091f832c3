"""Timing of the backward kernels at BASELINE config 2 (160^3 x 32, batch 1..B): HIP events, ms per call."""
import json
import sys

import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import neurite_amd as ne
from neurite_amd import synth


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    dev = torch.device('cuda:0')
    mov, fix, flow = synth.cfg2_batch(B, size=size, device=dev)
    res = {'batch': B, 'size': size}
    st = ne.layers.SpatialTransformer()
    dice = ne.metrics.Dice(check_input_limits=False)
    cce = ne.losses.CategoricalCrossentropy()
    nvox = B * size ** 3

    flow_g = flow.clone().requires_grad_()
    mov_g = mov.clone().requires_grad_()
    out = st([mov, flow_g])
    g = torch.randn_like(out)
    res['warp_fwd_ms'] = timeit(lambda: ne.deferred.materialize(st([mov, flow])))
    res['warp_bwd_flow_ms'] = timeit(lambda: torch.autograd.grad(out, flow_g, g, retain_graph=True))
    out2 = st([mov_g, flow_g])
    res['warp_bwd_flow_vol_ms'] = timeit(lambda: torch.autograd.grad(out2, [mov_g, flow_g], g, retain_graph=True))
    w = out.detach().clone().requires_grad_()
    d = dice.mean_dice(fix, w)
    res['dice_fwd_ms'] = timeit(lambda: dice.mean_dice(fix, out.detach()))
    res['dice_bwd_ms'] = timeit(lambda: torch.autograd.grad(d, w, retain_graph=True))
    pw = (out.detach() + 0.01).requires_grad_()
    c = cce(fix, pw)
    res['cce_fwd_ms'] = timeit(lambda: cce(fix, pw.detach()))
    res['cce_bwd_ms'] = timeit(lambda: torch.autograd.grad(c, pw, retain_graph=True))

    def step():
        f = flow.clone().requires_grad_()
        l = -dice.mean_dice(fix, st([mov, f]))
        l.backward()
    res['reg_step_fwd_bwd_ms'] = timeit(step)
    res['Mvox_per_s_fwd_bwd'] = nvox / res['reg_step_fwd_bwd_ms'] / 1e3

    fg = flow.clone().requires_grad_()
    df = ne.fused.warp_dice(mov, fg, fix)
    gd = torch.full_like(df, -1.0 / df.numel())
    res['fused_fwd_ms'] = timeit(lambda: ne.fused.warp_dice(mov, flow, fix))
    res['fused_bwd_ms'] = timeit(lambda: torch.autograd.grad(df, fg, gd, retain_graph=True))

    def fstep():
        f = flow.clone().requires_grad_()
        l = -ne.fused.warp_dice(mov, f, fix).mean()
        l.backward()
    res['fused_step_fwd_bwd_ms'] = timeit(fstep)
    res['fused_Mvox_per_s_fwd_bwd'] = nvox / res['fused_step_fwd_bwd_ms'] / 1e3
    print(json.dumps(res))


if __name__ == '__main__':
    main()
