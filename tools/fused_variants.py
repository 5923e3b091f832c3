#!/usr/bin/env python3
"""Macro variants of neurite_amd/csrc/fused.hip as separate libraries (tools/lab/libnrt_fused_<k>.so = the product's objects with
fused.o rebuilt under -DNRT_FUSED_EXP=<k>), timed through bench.py with NEURITE_AMD_LIB pointing at each.
    python tools/fused_variants.py --build          (here, no GPU)
    python tools/fused_variants.py                  (GPU box)"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, 'tools', 'lab')
LIBDIR = os.path.join(ROOT, 'neurite_amd', 'lib')
VARIANTS = os.environ.get('FUSED_VARIANTS', 'EXP=0 EXP=1 EXP=2 EXP=3 EXP=4').split()        # NAME=value macros NRT_FUSED_<NAME>
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-unused-function', '-Wno-pass-failed']

if '--build' in sys.argv:
    objs = [o for o in sorted(glob.glob(os.path.join(LIBDIR, '*.o'))) if os.path.basename(o) != 'fused.o']
    for k in VARIANTS:
        o = os.path.join(LAB, 'fused_exp%s.o' % k.replace('=', '').replace(',', '_'))
        subprocess.check_call(['hipcc'] + FLAGS + ['-DNRT_FUSED_%s' % m for m in k.split(',')] + ['-c', os.path.join(ROOT, 'neurite_amd', 'csrc', 'fused.hip'), '-o', o])
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [o, '-o', os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_'))])
    print('built', VARIANTS)
    sys.exit(0)
for k in VARIANTS:
    env = dict(os.environ, NEURITE_AMD_LIB=os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_')))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--warmup', '10', '--no-cpu-baseline', '--no-batch1', '--no-unet'],
                       env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(json.dumps({'variant': k, 'Mvox_s': j['value'], 'ms_per_step': j['ms_per_step'], 'kernel_ms': j['roofline']['avg_launch_ms']}), flush=True)
    except Exception as e:      # noqa
        print(json.dumps({'variant': k, 'error': (r.stderr or str(e))[-300:]}), flush=True)
