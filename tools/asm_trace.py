#!/usr/bin/env python3
"""Condensed view of a kernel's ISA: memory / sync instructions with VALU counts in between.
usage: asm_trace.py file.s kernel_substring [first_label last_label]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*%s\S*):' % re.escape(key), s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
lo = sys.argv[3] if len(sys.argv) > 3 else None
hi = sys.argv[4] if len(sys.argv) > 4 else None
on = lo is None
v = 0
for l in body:
    t = l.strip()
    if lo and t.startswith(lo + ':'): on = True
    if hi and t.startswith(hi + ':'): break
    if not on: continue
    if re.match(r'v_', t): v += 1; continue
    if re.match(r'(ds_|buffer_|global_|scratch_|s_waitcnt|s_barrier|\.LBB|s_cbranch|s_branch|s_nop|s_memtime)', t):
        if v: print('   ...%d valu' % v); v = 0
        print(t.split(';')[0][:100])
