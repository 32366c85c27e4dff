#!/usr/bin/env python3
"""Static vector-instruction counts of one kernel by SOURCE FUNCTION (the innermost function of leon_kernels.h a line belongs to):
    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only -o /tmp/leon_g.s csrc/leon_hip.cpp
    python tools/probe/valu_by_source.py /tmp/leon_g.s k_recon_displayILi3ELb0ELb0E [--lines]
Static, not executed counts: the reconstruction kernels are nearly straight-line code (their loops are unrolled), so the shares are
close to what a wave executes; loops that remain (the liveness scan) count once."""
import collections
import os
import re
import sys

asm, kern = sys.argv[1], sys.argv[2]
by_line = "--lines" in sys.argv
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(asm).read()
files = {}
for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
    files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
m = re.search(r'^(_ZN4leon\w*%s\w*):' % re.escape(kern), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
# function starts of leon_kernels.h: "name(" at the start of a definition line
src = open(os.path.join(root, "mpeg1video-decoder-webgl_amd", "csrc", "leon_kernels.h")).read().splitlines()
starts = []
for i, l in enumerate(src, 1):
    mm = re.match(r'^(?:template.*>\s*)?(?:__device__|__global__|static|inline|__host__).*?\b(\w+)\s*\(', l)
    if mm and not l.strip().endswith(';'):
        starts.append((i, mm.group(1)))
def func_of(line):
    name = "?"
    for i, n in starts:
        if i <= line:
            name = n
        else:
            break
    return name
cur = ("?", 0)
valu = collections.Counter(); other = collections.Counter()
for l in body.splitlines():
    l = l.strip()
    mm = re.match(r'\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (os.path.basename(files.get(int(mm.group(1)), "?")), int(mm.group(2)))
        continue
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    key = ("%s:%d" % cur) if by_line else (func_of(cur[1]) if cur[0] == "leon_kernels.h" else cur[0])
    (valu if op.startswith('v_') else other)[key] += 1
tot = sum(valu.values())
print("%s: %d vector instructions (static), %d others" % (m.group(1)[:60], tot, sum(other.values())))
for k, v in valu.most_common(40):
    print("  %-28s %5d  %4.1f %%   (+%d scalar / memory / LDS)" % (k, v, 100.0 * v / tot, other[k]))
