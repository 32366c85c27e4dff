#!/usr/bin/env python3
"""static VALU / SALU / memory instruction counts of the reconstruction kernels in an assembly listing
(hipcc -S --cuda-device-only): python tools/probe/static_count.py /tmp/leon_hip.s"""
import re, sys, collections
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN4leon\w+):', s, re.M):
    name = m.group(1)
    if 'k_recon' not in name and 'k_vlc' not in name: continue
    j = s.index('.Lfunc_end', m.end())
    v = sc = mem = 0
    for l in s[m.end():j].splitlines():
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'): continue
        op = l.split()[0]
        if op.startswith('v_'): v += 1
        elif op.startswith('s_'): sc += 1
        else: mem += 1
    short = re.sub(r'_ZN4leon\d+', '', name)[:40]
    print('%-42s valu %5d salu %5d mem %4d' % (short, v, sc, mem))
