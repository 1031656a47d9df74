"""Static instruction mix of the loops of one kernel (gfx950 ISA text from `hipcc -S --cuda-device-only`).
usage: isa_loops.py file.s <mangled-name-substring>  -- per backward branch: [label, first line, last line] and the counts of
MFMA / VALU / SALU / DS / VMEM instructions between the target label and the branch (both sides of inner branches counted)."""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r'^_Z\w+:', l) and key in l)
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
lines = text[start:end + 1]
labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
seen = set()
for i, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if not m or m.group(1) not in labels or labels[m.group(1)] >= i or m.group(1) in seen:
        continue
    seen.add(m.group(1))
    body = lines[labels[m.group(1)]:i + 1]
    c = Counter()
    for x in body:
        x = x.strip()
        if not x or x[0] in ';.' or x.endswith(':'):
            continue
        op = x.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'): c['wait'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'): c['vmem'] += 1
        else: c['other'] += 1
    print(m.group(1), labels[m.group(1)], i, dict(c), 'total', sum(c.values()))
