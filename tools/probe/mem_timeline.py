#!/usr/bin/env python3
"""The memory instructions, waits, barriers and branches of one kernel in program order, with the number of vector instructions
between them -- where a wave requests, where it waits, and what lies between (loads return in order: an `s_waitcnt vmcnt(0)` waits
for EVERYTHING requested before it):
    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/leon.s csrc/leon_hip.cpp
    python tools/probe/mem_timeline.py /tmp/leon.s k_recon_displayILi3ELb0ELb0E [max lines]"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_ZN4leon\w*%s\w*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
valu = 0
out = []
for l in body.splitlines():
    l = l.strip()
    if not l or (l.startswith((';', '.')) and not l.startswith('.LBB')):
        continue
    op = l.split()[0]
    if op.startswith('v_'):
        valu += 1
        continue
    if op.startswith(('buffer_', 'global_', 's_waitcnt', 's_barrier', 's_setprio', 'ds_bpermute', 's_cbranch', 's_endpgm', '.LBB', 'ds_', 's_load', 's_buffer_load')):
        key = 'ds' if op.startswith('ds_') and not op.startswith('ds_bpermute') else l.split(';')[0].strip()[:70]
        if out and out[-1][1] == key and valu == 0 and key == 'ds':
            out[-1][2] += 1
        else:
            out.append([valu, key, 1])
        valu = 0
for v, k, n in out[:limit]:
    print("%4d  %s%s" % (v, k, " x%d" % n if n > 1 else ""))
