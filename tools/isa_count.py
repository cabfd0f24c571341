#!/usr/bin/env python3
"""Count instructions per kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for n in re.findall(r'^(_Z\w+):', s, re.M):
    if pat not in n:
        continue
    i = s.index('\n' + n + ':')
    j = s.index('s_endpgm', i)
    ins = []
    for l in s[i:j].split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    k2 = s.find('.name:           ' + n)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', s[k2:k2 + 3000]) if k2 >= 0 else None
    sg = re.search(r'\.sgpr_count:\s+(\d+)', s[k2:k2 + 3000]) if k2 >= 0 else None
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', s[k2 - 1500:k2 + 3000]) if k2 >= 0 else None
    print(n, 'total', len(ins), 'valu', valu, 'salu', sum(v for k, v in c.items() if k.startswith('s_')),
          'ds', sum(v for k, v in c.items() if k.startswith('ds_')), 'vgpr', vg and vg.group(1), 'sgpr', sg and sg.group(1),
          'lds', lds and lds.group(1))
    print('   ', c.most_common(16))
