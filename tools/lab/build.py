"""Build tools/lab/liblab.so: kernels that were measured and did NOT become the product path (the LDS row cache), plus the
memory-system probes.  Lab code includes the product's headers but nothing in neurite_amd/ depends on it.

    python tools/lab/build.py [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(HERE, 'csrc')
PROD = os.path.join(ROOT, 'neurite_amd', 'csrc')
OUT = os.path.join(HERE, 'liblab.so')
LIB_SOURCES = ['gather_lc.hip', 'lab_api.hip', 'membench.hip']
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-unused-function',
         '-Wno-pass-failed', '-I' + PROD]
PROBES = ['occ_probe', 'dma_probe', 'valu_rate', 'valu_ops', 'vmem_issue', 'mix_probe', 'mfma_rate']      # stand-alone programs


def build(force=False, probes=False):
    srcs = [os.path.join(CSRC, f) for f in LIB_SOURCES]
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(PROD, '*.h')) + [os.path.abspath(__file__)]
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in deps):
        objs, procs = [], []
        for s in srcs:
            o = os.path.join(HERE, os.path.basename(s)[:-4] + '.o')
            objs.append(o)
            procs.append(subprocess.Popen(['hipcc'] + FLAGS + ['-c', s, '-o', o]))
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError('hipcc failed')
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT], check=True)
    if probes:
        for name in PROBES:
            exe = os.path.join(HERE, name)
            src = os.path.join(CSRC, name + '.hip')
            if force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
                subprocess.run(['hipcc', '-O2', '--offload-arch=gfx950', src, '-o', exe], check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, probes='--probes' in sys.argv))
