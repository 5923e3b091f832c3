#!/usr/bin/env python3
"""Scan the gfx950 ISA of the product kernels for patterns that hid large factors in round 3:
loops that contain integer division sequences (v_rcp_iflag / 64-bit emulation), and loops with many full waits
(s_waitcnt vmcnt(0) / lgkmcnt(0)) -- dependent memory round trips.   usage: isa_scan.py [file.hip ...] (default: all of csrc/)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'neurite_amd', 'csrc', '*.hip')))
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-unused-function', '-Wno-pass-failed',
         '--offload-device-only', '-S']
for f in files:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        r = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [f, '-o', out], capture_output=True, text=True)
        if r.returncode:
            print(f, 'COMPILE FAILED', r.stderr[-300:])
            continue
        s = open(out).read()
    for m in re.finditer(r'^(\S+):\s*; @\S+\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if '.amdhsa_kernel' not in body:
            continue
        # split into basic blocks; a block is "in a loop" if its header comment says so
        blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', body)
        div = full_v = full_l = 0
        for b in blocks:
            head = b[:300]
            if 'Loop' not in head:
                continue
            div += len(re.findall(r'v_rcp_iflag_f32|v_rcp_f32', b))
            full_v += len(re.findall(r's_waitcnt vmcnt\(0\)', b))
            full_l += len(re.findall(r's_waitcnt lgkmcnt\(0\)', b))
        if div >= 2 or full_v >= 6 or full_l >= 12:
            short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:70]
            print('%-22s %-72s divisions-in-loops %3d  vmcnt(0)-in-loops %3d  lgkmcnt(0)-in-loops %3d' % (os.path.basename(f), short, div, full_v, full_l))
