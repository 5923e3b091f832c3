#!/usr/bin/env python3
"""A/B harness for working copies of neurite_amd/csrc/fused.hip + fused_wc.h: each variant is a separate library
(tools/lab/libnrt_fused_<k>.so = the product's objects with fused.o rebuilt under -DNRT_FUSED_<NAME>=<value> ...), timed through
bench.py with NEURITE_AMD_LIB pointing at it.  The product sources carry NO such switches (round 5): a lab session adds its
`#if NRT_FUSED_<NAME>` blocks to a working copy, measures, and removes them before the commit -- the probe builds behind
profiles/r04_lab/l1_access_curve.jsonl (NRT_FUSED_EXP) are in the history at commit e0b416e, those behind
profiles/r05_lab/wc_probes.jsonl were never committed with the sources.
    FUSED_VARIANTS="A=1 A=1,B=2" python tools/fused_variants.py --build          (here, no GPU)
    FUSED_VARIANTS="A=1 A=1,B=2" FUSED_REPS=3 python tools/fused_variants.py      (GPU box; variants alternate REPS times)
    FUSED_FETCH=1 ...                              also one rocprofv3 --pmc FETCH_SIZE pass per variant over tools/fused_small.py (batch
                                                   FUSED_BATCH, default 4): HBM-side read bytes per launch of the gather (x2-corrected)"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, 'tools', 'lab')
LIBDIR = os.path.join(ROOT, 'neurite_amd', 'lib')
VARIANTS = os.environ.get('FUSED_VARIANTS', 'BASE=0').split()        # NAME=value macros NRT_FUSED_<NAME>
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-unused-function', '-Wno-pass-failed']

if '--build' in sys.argv:
    objs = [o for o in sorted(glob.glob(os.path.join(LIBDIR, '*.o'))) if os.path.basename(o) != 'fused.o']
    for k in VARIANTS:
        o = os.path.join(LAB, 'fused_exp%s.o' % k.replace('=', '').replace(',', '_'))
        subprocess.check_call(['hipcc'] + FLAGS + ['-DNRT_FUSED_%s' % m for m in k.split(',')] + ['-c', os.path.join(ROOT, 'neurite_amd', 'csrc', 'fused.hip'), '-o', o])
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [o, '-o', os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_'))])
    print('built', VARIANTS)
    sys.exit(0)
def fetch_pass(k, lib):
    """mean FETCH_SIZE (KiB, raw) per dispatch of the warp_dice kernels under this variant's library"""
    import csv
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix='fv_', dir='/tmp')
    env = dict(os.environ, NEURITE_AMD_LIB=lib, TMPDIR='/tmp', PYTHONPATH=ROOT)
    subprocess.run(['rocprofv3', '--pmc', 'FETCH_SIZE', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
                    sys.executable, os.path.join(ROOT, 'tools', 'fused_small.py'), os.environ.get('FUSED_BATCH', '4')],
                   env=env, cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    vals = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == 'FETCH_SIZE' and 'warp_dice' in row.get('Kernel_Name', ''):
                vals.setdefault(row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60], []).append(float(row['Counter_Value']))
    shutil.rmtree(d, ignore_errors=True)
    return {kn: round(2 * 1024 * sum(v) / len(v) / 1e9, 4) for kn, v in vals.items()}


REPS = int(os.environ.get('FUSED_REPS', '1'))
STEPS = os.environ.get('FUSED_STEPS', '40')
if os.environ.get('FUSED_FETCH'):
    for k in VARIANTS:
        try:
            print(json.dumps({'variant': k, 'fetch_GB_per_launch_x2_corrected': fetch_pass(k, os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_')))}), flush=True)
        except Exception as e:      # noqa
            print(json.dumps({'variant': k, 'fetch_error': str(e)[-300:]}), flush=True)
for k in VARIANTS * REPS:
    env = dict(os.environ, NEURITE_AMD_LIB=os.path.join(LAB, 'libnrt_fused_%s.so' % k.replace('=', '').replace(',', '_')))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', STEPS, '--warmup', '10', '--no-cpu-baseline', '--no-batch1', '--no-unet'],
                       env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(json.dumps({'variant': k, 'Mvox_s': j['value'], 'ms_per_step': j['ms_per_step'], 'kernel_ms': j['roofline']['avg_launch_ms'],
                          'isolated_kernel_ms': j['roofline'].get('isolated_launch', {}).get('avg_launch_ms'),
                          'standalone_interpn_ms': j['roofline'].get('standalone_interpn', {}).get('avg_launch_ms')}), flush=True)
    except Exception as e:      # noqa
        print(json.dumps({'variant': k, 'error': (r.stderr or str(e))[-300:]}), flush=True)
