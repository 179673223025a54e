#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying basic blocks of a kernel in a hipcc -S listing.
usage: tools/asm_blocks.py file.s <substring of mangled kernel name>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'\n(_Z\S*' + re.escape(pat) + r'\S*):', s):
    sym = m.group(1)
    i = m.end()
    j = s.index('s_endpgm', i)
    lines = s[i:j].splitlines()
    blocks, cur = [], []
    for l in lines:
        t = l.strip()
        if re.match(r'\.LBB\d+_\d+:', t):
            blocks.append(cur); cur = []
        elif t and not t.startswith(('.', ';', '//')):
            cur.append(t.split()[0])
            if t.startswith(('s_cbranch', 's_branch')):
                blocks.append(cur); cur = []
    blocks.append(cur)
    print(sym[:90])
    for b in blocks:
        c = collections.Counter(b)
        nm = sum(v for k, v in c.items() if 'mfma' in k)
        if nm >= 8:
            print(f'  block: {len(b)} instrs, {nm} mfma | ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(16)))
