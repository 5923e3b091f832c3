#!/usr/bin/env python3
"""Compact instruction-class trace of one kernel from a hipcc -save-temps .s file.
usage: isa_trace.py file.s kernel-name-substring [max chars]
M mfma, r/w ds_read/ds_write, G/S global load/store, [..] s_waitcnt, |B| barrier, v/s other VALU/SALU."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*%s\S*):' % re.escape(pat), s, re.M)
if not m:
    sys.exit('no kernel label matching ' + pat)
i = m.start()
j = s.find('.end_amdhsa_kernel', i)
out = []
for l in s[i:j].split('\n'):
    l = l.strip()
    if not l or l[0] in ';.':
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): c = 'M'
    elif op.startswith('ds_read'): c = 'r'
    elif op.startswith('ds_write'): c = 'w'
    elif op.startswith(('global_load', 'buffer_load')): c = 'G'
    elif op.startswith(('global_store', 'buffer_store')): c = 'S'
    elif op.startswith('s_waitcnt'): c = '[' + l.split(None, 1)[1].replace('vmcnt', 'v').replace('lgkmcnt', 'l') + ']'
    elif op.startswith('s_barrier'): c = '|B|'
    elif op.startswith(('s_cbranch', 's_branch')): c = '<br>'
    elif op.endswith(':'): c = '\n' + op
    elif op.startswith('v_'): c = 'v'
    elif op.startswith('s_'): c = 's'
    else: c = '?'
    out.append(c)
txt = ''.join(out)
print(txt[:int(sys.argv[3]) if len(sys.argv) > 3 else 20000])
