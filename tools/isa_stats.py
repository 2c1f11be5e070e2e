#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file.  usage: isa_stats.py file.s substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
funcs = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(funcs), 2):
    name, body = funcs[i], funcs[i + 1]
    if want not in name:
        continue
    body = body[:body.find('.Lfunc_end')]
    ins = [l.strip().split()[0] for l in body.split('\n')
           if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    salu = sum(v for k, v in c.items() if k.startswith('s_'))
    print(name, 'total', len(ins), 'valu', valu, 'salu', salu)
    for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
        print('  %-28s %d' % (k, v))
