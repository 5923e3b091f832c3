"""
torch-CPU vectorised restatement of interpn linear (TEST INFRASTRUCTURE -- see oracle/__init__.py).

SURVEY.md 8(d) asks for the CPU reference path of BASELINE config 1 in two forms: the op-for-op NumPy restatement on one
thread (oracle/np_oracle.py) and "a torch-CPU vectorised version on all host cores".  This is the latter: the same op
sequence as neurite/tf/utils/utils.py:137-191 (floor, clips, int cast, weights l1 - clipped and 1 - w, corners in
itertools.product order, ((w0*w1)*w2) * row accumulated in that order) written with torch CPU ops in float32, so it is
bit-identical to np_oracle.interpn (checked in tests/test_oracle.py and again inside bench.py's cpu_baseline leg).
Used only by bench.py's `cpu_baseline` and tests.
"""

import itertools

import torch


def interpn_linear(vol, loc):
    """vol [*S] or [*S, C] float32 CPU tensor, loc [*S', D] float32.  Returns [*S'] or [*S', C]."""
    D = loc.shape[-1]
    squeeze = vol.dim() == D
    if squeeze:
        vol = vol[..., None]                                               # :119-120
    S = vol.shape[:-1]
    flat = vol.reshape(-1, vol.shape[-1])                                  # :177
    loc0 = torch.floor(loc)                                                # :139
    i_lo, i_hi, w_lo, w_hi = [], [], [], []
    for d in range(D):
        mx = float(S[d] - 1)
        cl = loc[..., d].clamp(0.0, mx)                                    # :142
        l0 = loc0[..., d].clamp(0.0, mx)                                   # :143
        l1 = (l0 + 1.0).clamp(0.0, mx)                                     # :146
        i_lo.append(l0.to(torch.int64)); i_hi.append(l1.to(torch.int64))   # :147
        w1 = l1 - cl                                                       # :152 weight of the lower corner
        w_lo.append(w1); w_hi.append(1.0 - w1)                             # :153
    out = None
    for c in itertools.product([0, 1], repeat=D):                          # :159-191
        idx = None
        wt = None
        for d in range(D):
            sub = i_hi[d] if c[d] else i_lo[d]
            idx = sub if idx is None else idx * S[d] + sub                 # sub2ind2d: row-major
            w = w_hi[d] if c[d] else w_lo[d]
            wt = w if wt is None else wt * w                               # prod_n: left to right
        term = wt[..., None] * flat[idx.reshape(-1)].reshape(idx.shape + (flat.shape[-1],))
        out = (term + 0.0) if out is None else out + term                # :160 interp_vol = 0, then += (0 + x: -0 becomes +0)
    return out[..., 0] if squeeze else out
