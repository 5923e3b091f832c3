"""
Synthetic workloads of the BASELINE.json configurations, generated with torch on
whatever device is asked for.  Used by bench.py, __graft_entry__.smoke() and the tests; not part of
the hot path.
"""

import numpy as np
import torch
import torch.nn.functional as F


def _smooth_field(rng, channels, size, coarse, device):
    """N(0,1) noise on a coarse^3 grid, tri-linearly upsampled to size^3: [channels, size, size, size]."""
    lo = torch.from_numpy(rng.standard_normal((1, channels, coarse, coarse, coarse)).astype(np.float32)).to(device)
    return F.interpolate(lo, size=(size, size, size), mode='trilinear', align_corners=True)[0]


def blob_labels(seed, size=160, nb_labels=32, coarse=10, device='cpu'):
    """Blob-like label map [size^3] int64: arg-max of nb_labels smooth random fields."""
    rng = np.random.default_rng(seed)
    return torch.argmax(_smooth_field(rng, nb_labels, size, coarse, device), 0)


def one_hot_volume(seed, size=160, nb_labels=32, device='cpu'):
    """[size, size, size, nb_labels] float32 one-hot of blob_labels(seed)."""
    lab = blob_labels(seed, size, nb_labels, device=device)
    out = torch.zeros((size, size, size, nb_labels), dtype=torch.float32, device=device)
    out.scatter_(-1, lab.unsqueeze(-1), 1.0)
    return out


def smooth_displacement(seed, size=160, sigma=3.0, coarse=20, device='cpu'):
    """Smooth displacement field [size^3, 3] float32 in voxel units with std ~= sigma."""
    rng = np.random.default_rng(seed)
    f = _smooth_field(rng, 3, size, min(coarse, size), device)
    f = f * (sigma / float(f.std()))
    return f.permute(1, 2, 3, 0).contiguous()


def rough_displacement(seed, size=160, amplitude=80.0, device='cpu'):
    """Worst case for the gather: i.i.d. U(-amplitude, amplitude) per voxel (incoherent reads)."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(-amplitude, amplitude, (size, size, size, 3)).astype(np.float32)
    return torch.from_numpy(f).to(device)


def cfg2_batch(batch, size=160, nb_labels=32, device='cpu', seed0=100, rough=False):
    """
    BASELINE config 2 / 4: moving, fixed [B, S, S, S, L] one-hot float32 and trf [B, S, S, S, 3].
    Seeds follow SURVEY.md section 8d (volume b uses seeds seed0 + 3b .. seed0 + 3b + 2).
    """
    mov, fix, trf = [], [], []
    for b in range(batch):
        s = seed0 + 3 * b
        mov.append(one_hot_volume(s, size, nb_labels, device))
        fix.append(one_hot_volume(s + 1, size, nb_labels, device))
        trf.append(rough_displacement(s + 2, size, device=device) if rough
                   else smooth_displacement(s + 2, size, device=device))
    return torch.stack(mov), torch.stack(fix), torch.stack(trf)
