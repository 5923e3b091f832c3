#!/usr/bin/env python3
"""Element-wise error of the Conv3D kernels against float64 (GPU session tool): for every variant the largest relative error over the
outputs with |ref| >= floor * max|ref|, the largest absolute error below that floor in units of max|ref|, and what the classical
bound gamma_K * sum|x_i w_i| predicts.   python tools/conv_err_report.py"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurite_amd import models as nm          # noqa: E402

dev = torch.device('cuda:0')


def stats(got, ref, absref, K, floor=1e-3):
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    mx = float(ref.abs().max())
    big = ref.abs() >= floor * mx
    rel = float((err[big] / ref.abs()[big]).max())
    small = float(err[~big].max() / mx) if (~big).any() else 0.0
    # error relative to the condition of each output: |err| / sum |x_i w_i|  (what fp32 accumulation can promise: ~ K^0.5 .. K times 2^-24)
    cond = float((err / absref.double().clamp_min(1e-30)).max())
    return {'K': K, 'max_rel_err_where_ref_ge_1e-3max': rel, 'max_abs_err_below_floor_over_max': small, 'max_err_over_sum_abs_terms': cond,
            'frac_below_floor': float((~big).double().mean())}


def ref_conv(x, w, b, dil=1):
    # x [B, X, Y, Z, C] -> float64 reference of the SAME cross-correlation + the sum of |terms| (the condition of each output)
    xt = x.double().permute(0, 4, 1, 2, 3).cpu()
    wt = w.double().permute(4, 3, 0, 1, 2).cpu()
    k = w.shape[0]
    pad = ((k - 1) * dil) // 2
    y = TF.conv3d(xt, wt, b.double().cpu(), padding=pad, dilation=dil)
    ya = TF.conv3d(xt.abs(), wt.abs(), b.double().abs().cpu(), padding=pad, dilation=dil)
    return y.permute(0, 2, 3, 4, 1), ya.permute(0, 2, 3, 4, 1)


rng = np.random.default_rng(0)
rows = []
for (cin, cout, S, variants) in ((16, 32, (16, 16, 32), (2, 5, 1)), (32, 64, (12, 12, 16), (2, 5)), (96, 32, (12, 12, 16), (2, 5)),
                                 (48, 16, (16, 16, 32), (2, 5)), (1, 16, (16, 16, 32), (0,))):
    conv = nm._Conv('c', cin, cout, (3, 3, 3), 1, 'same', None).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32))
    b = torch.from_numpy((rng.standard_normal(cout) * 0.1).astype(np.float32))
    with torch.no_grad():
        conv.kernel.copy_(w); conv.bias.copy_(b)
    x = torch.from_numpy(rng.standard_normal((1,) + S + (cin,)).astype(np.float32))
    ref, absref = ref_conv(x, w, b)
    for v in variants:
        y = conv(x.to(dev), variant=v).cpu()
        r = stats(y, ref, absref, 27 * cin)
        r.update(kernel='conv3d variant %d' % v, cin=cin, cout=cout)
        rows.append(r); print(json.dumps(r), flush=True)
# the folded decoder form
for (c0, c1, cout, S) in ((16, 32, 16, (16, 16, 32)), (32, 64, 32, (8, 12, 16))):
    conv = nm._Conv('c', c0 + c1, cout, (3, 3, 3), 1, 'same', None).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, c0 + c1, cout)) / np.sqrt(27 * (c0 + c1))).astype(np.float32))
    b = torch.from_numpy((rng.standard_normal(cout) * 0.1).astype(np.float32))
    with torch.no_grad():
        conv.kernel.copy_(w); conv.bias.copy_(b)
    skip = torch.from_numpy(rng.standard_normal((1,) + S + (c0,)).astype(np.float32))
    lo = torch.from_numpy(rng.standard_normal((1,) + tuple(s // 2 for s in S) + (c1,)).astype(np.float32))
    cat = torch.cat([skip, lo.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)], -1)
    ref, absref = ref_conv(cat, w, b)
    for v in (4, 2):
        y = conv(skip.to(dev), lo=lo.to(dev), up=(2, 2, 2), variant=v).cpu()
        r = stats(y, ref, absref, 27 * (c0 + c1))
        r.update(kernel='decoder conv variant %d (%s)' % (v, 'folded 8 taps' if v == 4 else '27 taps'), cin=c0 + c1, cout=cout)
        rows.append(r); print(json.dumps(r), flush=True)
