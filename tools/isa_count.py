#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -save-temps .s file: python tools/isa_count.py file.s [name-substring]
(whole kernel body; loops are counted once)."""
import collections
import re
import sys


def count(path, sub="k_winILb0E"):
    s = open(path).read()
    m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(sub), s, re.S | re.M)
    body = m.group(2).split('\n')
    ins = []
    for ln in body:
        t = ln.strip()
        if not ln.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t)
    c = collections.Counter()
    for t in ins:
        op = t.split()[0]
        if op.startswith('v_'):
            c['VALU'] += 1
        elif op.startswith('s_'):
            c['SALU/ctl'] += 1
        elif op.startswith('ds_'):
            c['LDS'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            c['VMEM'] += 1
        else:
            c['other'] += 1
    for key, pat in (('dpp', '_dpp'), ('v_cmp', 'v_cmp'), ('v_cndmask', 'v_cndmask'), ('v_max3', 'v_max3'),
                     ('v_readlane', 'v_readlane'), ('s_nop', 's_nop'), ('s_waitcnt', 's_waitcnt'), ('s_barrier', 's_barrier')):
        c[key] = sum(1 for t in ins if pat in t.split()[0] or (pat == '_dpp' and '_dpp' in t))
    return m.group(1), len(ins), dict(c)


if __name__ == "__main__":
    print(*count(sys.argv[1], *(sys.argv[2:3])))
