#!/usr/bin/env python3
"""Counts, per kernel of an assembly listing (hipcc -S --cuda-device-only), the v_bitop3_b32 a ^ (b & m) instructions by
the register-file pattern of their sources: on gfx950 three VGPR sources of the SAME index parity cost ~4.5 cycles per
wave-instruction and SIMD, any other split ~2.8 (tools/micro/vgpr_bank.hip, profiles/r2_vgpr_bank_sweep.txt).
usage: tools/isa_banks.py file.s"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_Z[^\n]*:\s*; @)', txt):
    name = f.split(':')[0]
    if not name.startswith('_Z'):
        continue
    ops = re.findall(r'v_bitop3_b32 v(\d+), (\S+), (\S+), (\S+) bitop3:0x78', f)
    if len(ops) < 200:
        continue
    c = collections.Counter()
    for d, a, b, m in ops:
        srcs = [x.rstrip(',') for x in (a, b, m)]
        vs = [int(x[1:]) & 1 for x in srcs if re.fullmatch(r'v\d+', x)]
        if len(vs) == 3:
            c['3v_same_parity' if len(set(vs)) == 1 else '3v_split'] += 1
        elif len(vs) == 2:
            c['2v+s_same' if len(set(vs)) == 1 else '2v+s_split'] += 1
        else:
            c['other'] += 1
    print(name[:70], len(ops), dict(c))
