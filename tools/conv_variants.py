#!/usr/bin/env python3
"""Time nrt_conv3d_up2_f32 (and the 27-tap kernel) at the two decoder shapes of BASELINE config 3 for several builds of
neurite_amd/csrc/conv.hip (macro variants compiled by `--build` HERE, timed on the GPU box).
    python tools/conv_variants.py --build            (no GPU needed)
    python tools/conv_variants.py                    (GPU)  -> one JSON line per variant and shape"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, 'tools', 'lab')
SRC = os.path.join(ROOT, 'neurite_amd', 'csrc', 'conv.hip')
VARIANTS = {
    'base': [],
    'wdist1': ['-DU2_WDIST=1'],
    'wdist3': ['-DU2_WDIST=3'],
    'defer2': ['-DU2_DEFER_MAXNT=2'],
    'defer0': ['-DU2_DEFER_MAXNT=0'],
}
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-unused-function',
         '-Wno-pass-failed', '-shared']


def build():
    procs = []
    for name, defs in VARIANTS.items():
        out = os.path.join(LAB, 'libconv_%s.so' % name)
        procs.append((name, subprocess.Popen(['hipcc'] + FLAGS + defs + [SRC, '-o', out])))
    for name, p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed for ' + name)
    print('built', ', '.join(VARIANTS))


def main():
    import torch
    dev = torch.device('cuda:0')
    shapes = [(16, 32, 16, 160), (32, 64, 32, 80)]
    names = [a for a in sys.argv[1:] if not a.startswith('-')] or list(VARIANTS)
    for name in names:
        lib = ctypes.CDLL(os.path.join(LAB, 'libconv_%s.so' % name))
        lib.nrt_conv3d_up2_packed_weight_floats.restype = ctypes.c_size_t
        for (c0, c1, cout, S) in shapes:
            torch.manual_seed(1)
            skip = torch.randn(1, S, S, S, c0, device=dev)
            lo = torch.randn(1, S // 2, S // 2, S // 2, c1, device=dev)
            w = torch.randn(3, 3, 3, c0 + c1, cout, device=dev) * 0.03
            bias = torch.zeros(cout, device=dev)
            out = torch.empty(1, S, S, S, cout, device=dev)
            n = lib.nrt_conv3d_up2_packed_weight_floats(c0, c1, cout)
            packed = torch.empty(n, device=dev)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            shp = (ctypes.c_int * 3)(S, S, S)
            assert lib.nrt_conv3d_up2_pack_weights_f32(P(w), c0, c1, cout, P(packed), st) == 0
            run = lambda: lib.nrt_conv3d_up2_f32(P(skip), c0, P(lo), c1, P(packed), P(bias), P(out), 1, shp, cout, 1, st)
            for _ in range(3):
                assert run() == 0
            torch.cuda.synchronize()
            best = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best.append(e0.elapsed_time(e1) / 10)
            gf = 2.0 * S ** 3 * (27 * c0 + 8 * c1) * cout / 1e9
            ms = sorted(best)[len(best) // 2]
            print(json.dumps({'variant': name, 'c0': c0, 'c1': c1, 'cout': cout, 'size': S, 'ms': round(ms, 4), 'ms_min': round(min(best), 4),
                              'frac_of_fp32_mfma_peak': round(gf / ms / 157.3, 4)}), flush=True)


if __name__ == '__main__':
    build() if '--build' in sys.argv else main()
