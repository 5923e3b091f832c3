"""
TEST INFRASTRUCTURE ONLY -- build tests/emu/_build/libemu_fused.so: neurite_amd/csrc/fused.hip compiled for the CPU.

The kernel sources are copied and patched in three places that cannot be expressed on a host target:
  * the two inline-asm helpers of nrt_common.h (v_mad_u32_u24, v_lshl_add_u32) get their C meaning,
  * `extern __shared__` (dynamic LDS) becomes `extern` (the arrays live in emu_fused.cpp),
  * the relative include of include/neurite_amd.h is made absolute.
Everything else -- kernels, lambdas, launch code, C entry points -- is compiled as it is.
"""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'neurite_amd', 'csrc')
BUILD = os.path.join(HERE, '_build')
CLANG = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')


def available():
    return os.path.exists(CLANG)


def build(force=False):
    out = os.path.join(BUILD, 'libemu_fused.so')
    srcs = [os.path.join(CSRC, f) for f in ('fused.hip', 'interpn_core.h', 'dice_reduce.h', 'nrt_common.h')]
    deps = srcs + [os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(HERE, 'emu_fused.cpp'), __file__]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    dst = os.path.join(BUILD, 'csrc')
    os.makedirs(dst, exist_ok=True)
    for s in srcs:
        text = open(s).read()
        if s.endswith('nrt_common.h'):
            text, n1 = re.subn(r'asm\("v_mad_u32_u24[^;]*;', 'r = (a & 0xffffffu) * (b & 0xffffffu) + c;', text)
            text, n2 = re.subn(r'asm\("v_lshl_add_u32[^;]*;', 'r = (q << 1) + q;', text)
            assert n1 == 1 and n2 == 1, 'nrt_common.h: inline-asm helpers not found'
            text = text.replace('#include "../../include/neurite_amd.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'neurite_amd.h'))
        text = text.replace('extern __shared__', 'extern')
        open(os.path.join(dst, os.path.basename(s)), 'w').write(text)
    cmd = [CLANG, '-std=c++20', '-O1', '-ffp-contract=off', '-pthread', '-fPIC', '-shared', '-Wno-everything',
           '-I', HERE, '-I', BUILD, os.path.join(HERE, 'emu_fused.cpp'), '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('emulator build failed:\n' + r.stderr[-4000:])
    return out


if __name__ == '__main__':
    print(build(force=True))
