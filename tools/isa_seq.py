#!/usr/bin/env python3
"""tools/isa_seq.py FILE.s KERNEL_SUBSTRING [tail]: instruction-class counts of one kernel of a `hipcc -S --cuda-device-only` listing and the ORDER of its memory
operations, waits and barriers (run-length compressed) - what the chain's and the conv's kernels were read with in rounds 5-6."""
import collections
import re
import sys

s = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(sys.argv[2]) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
body = s[start + 1:end]
c = collections.Counter()
for l in body:
    t = l.strip().split()
    if t and not t[0].startswith((';', '.')) and not t[0].endswith(':'):
        c[t[0]] += 1
print(s[start].split(':')[0], len(body), 'lines')
valu = sum(v for k, v in c.items() if k.startswith('v_'))
print('VALU', valu, ' SALU', sum(v for k, v in c.items() if k.startswith('s_')), ' total', sum(c.values()))
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    if re.match(r'(global_|flat_|buffer_|scratch_|ds_|s_barrier|s_waitcnt|s_load|s_cbranch)', k):
        print('  %-28s %d' % (k, v))
seq = []
for l in body:
    t = l.strip().split(';')[0].strip()
    if re.match(r'(global_|flat_|buffer_|scratch_|s_barrier|s_waitcnt|ds_|s_cbranch|s_branch)', t):
        seq.append(t.split()[0] + (' ' + t.split(None, 1)[1] if t.startswith('s_waitcnt') else ''))
out, prev, n = [], None, 0
for x in seq:
    if x == prev:
        n += 1
    else:
        if prev:
            out.append('%s x%d' % (prev, n))
        prev, n = x, 1
out.append('%s x%d' % (prev, n))
tail = int(sys.argv[3]) if len(sys.argv) > 3 else len(out)
print('\n'.join(out[-tail:]))
