"""
Build libneurite_amd.so (the C-ABI library declared in include/neurite_amd.h) for gfx950.

    python -m neurite_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is written in-tree (neurite_amd/lib/) so that it
travels with the source snapshot; it is git-ignored.
"""

import glob
import hashlib
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


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, '..', 'include', 'neurite_amd.h')]


def _sha(*chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else c.encode())
        h.update(b'\0')
    return h.hexdigest()


def _read(path):
    with open(path, 'rb') as f:
        return f.read()


_compiler = None


def compiler_version():
    """first line of `hipcc --version` that names the HIP / clang build.  Part of the build id: several kernels rely on
    hand-placed hazard padding and hand-counted waits around asm statements, which a different compiler may schedule differently -- a
    library built by another hipcc is a different library and says so."""
    global _compiler
    if _compiler is None:
        try:
            out = subprocess.run([os.environ.get('HIPCC', 'hipcc'), '--version'], capture_output=True, text=True, timeout=60).stdout
            lines = [ln.strip() for ln in out.splitlines() if 'version' in ln.lower()]
            _compiler = ' | '.join(lines[:2]) or 'unknown'
        except Exception:   # noqa
            _compiler = 'unknown'
    return _compiler


def _common_hash():
    """what every object depends on besides its own source: the headers, the compile flags and the compiler"""
    return _sha(' '.join(FLAGS), compiler_version(), *[_read(h) for h in _headers()])


def build_id():
    """Identity of the sources the library SHOULD be built from: sha256 over flags, headers and every .hip file (16 hex digits).
    The library carries the id it was built from (`nrt_build_id()`, a string constant in api.o); mtimes play no part -- a git checkout,
    a copied tree or a shipped binary newer than the sources it does not match are all told apart by content."""
    return _sha(_common_hash(), *[_read(s) for s in sources()])[:16]


# the translation units (with every header) the C = 32 gather kernels of the bench line are compiled from: fused warp + Dice, the stand-alone
# warp and the Dice reduction that follows them
GATHER_SOURCES = ('fused.hip', 'interpn.hip', 'dice.hip')


def gather_sources_id():
    """Identity of the sources of the bench line's kernels (16 hex digits over flags, compiler, headers and GATHER_SOURCES).
    profiles/hbm_traffic.json stamps every counter entry with it (and with the id of the whole library it was measured on); bench.py
    quotes `roofline.traffic` only while the loaded library still is the tree's build AND this id is the entry's -- a kernel edit that
    keeps the kernel's NAME no longer keeps its old bytes, while an edit of, say, lc3d.hip does not void them."""
    return _sha(_common_hash(), *[_read(os.path.join(CSRC, f)) for f in GATHER_SOURCES])[:16]


_ID_MARK = b'NRT_BUILD_ID='


def library_build_id(path=None):
    """the id embedded in an existing library (read from the file, nothing is loaded), or None"""
    path = path or LIB
    if not os.path.exists(path):
        return None
    blob = _read(path)
    k = blob.find(_ID_MARK)
    if k < 0:
        return None
    return blob[k + len(_ID_MARK):k + len(_ID_MARK) + 16].decode('ascii', 'replace')


def is_stale():
    return library_build_id() != build_id()


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ into one shared library.  Returns the library path."""
    bid = build_id()
    if not force and library_build_id() == bid:
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    common = _common_hash()
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        # an object is reused when the hash of what it was compiled from (source, headers, flags; for api.o also the library id) matches
        is_api = os.path.basename(src) == 'api.hip'
        want = _sha(common, _read(src), bid if is_api else '')
        stamp = obj + '.sha'
        if not force and os.path.exists(obj) and os.path.exists(stamp) and _read(stamp).decode() == want:
            continue
        cmd = [hipcc] + FLAGS + (['-DNRT_BUILD_ID_STRING="%s"' % bid] if is_api else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, stamp, want, subprocess.Popen(cmd)))
    for src, stamp, want, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + src)
        with open(stamp, 'w') as f:
            f.write(want)
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    got = library_build_id()
    if got != bid:
        raise RuntimeError('the library just built reports build id %r, expected %r' % (got, bid))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
