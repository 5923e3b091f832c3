#!/usr/bin/env python3
"""Compile one HIP source for gfx950 with the product flags (+ extra -D.. flags) and print registers / spills / scratch of the
kernels whose name contains a pattern.   usage: kernel_regs.py file.hip pattern [extra hipcc flags...]
The .s file stays in /tmp/kregs_<pid>/ for tools/isa_trace.py."""
import os
import re
import subprocess
import sys

src, pat, extra = os.path.abspath(sys.argv[1]), sys.argv[2], sys.argv[3:]
d = '/tmp/kregs_%d' % os.getpid()
os.makedirs(d, exist_ok=True)
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize',
       '-Wno-unused-function', '-Wno-pass-failed', '-save-temps=obj', '-c', src, '-o', os.path.join(d, 'x.o')] + extra
r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
sfile = [f for f in os.listdir(d) if f.endswith('gfx950.s')][0]
s = open(os.path.join(d, sfile)).read()
for blk in re.finditer(r'- \.agpr_count.*?\.wavefront_size', s, re.S):
    b = blk.group(0)
    n = re.search(r'\.name:\s+(\S+)', b).group(1)
    if pat in n:
        g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
        print('%-70s total %s agpr %s vspill %s sspill %s scratch %s' % (n[:70], g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'),
                                                                         g('sgpr_spill_count'), g('private_segment_fixed_size')))
print(os.path.join(d, sfile))
