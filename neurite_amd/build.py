"""
Build libneurite_amd.so (the C-ABI library declared in include/neurite_amd.h) for gfx950.

    python -m neurite_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is written in-tree (neurite_amd/lib/) so that it
travels with the source snapshot; it is git-ignored.
"""

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libneurite_amd.so')
ARCH = 'gfx950'

# -ffp-contract=off: the interpolation kernels reproduce the reference's float32 op sequence with one
# rounding per op; an FMA would change the bits.
# -fno-slp-vectorize: hipcc otherwise packs the corner blends into v_pk_*_f32 across rows, which needs
# register shuffles of values whose loads are still in flight and drains the load queue (s_waitcnt
# vmcnt(0)) in every loop iteration of the pipelined gather kernels.
FLAGS = ['-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
         '-Wall', '-Wno-unused-function', '-Wno-pass-failed']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _newest_input():
    files = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(HERE, '..', 'include', 'neurite_amd.h'), os.path.abspath(__file__)]
    return max(os.path.getmtime(f) for f in files)


def is_stale():
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < _newest_input()


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ into one shared library.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    # an object is reused when it is newer than its source, every header and this file (the flags)
    common = max(os.path.getmtime(f) for f in glob.glob(os.path.join(CSRC, '*.h')) +
                 [os.path.join(HERE, '..', 'include', 'neurite_amd.h'), os.path.abspath(__file__)])
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(common, os.path.getmtime(src)):
            continue
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + src)
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
