"""dev tool: instruction mix of one kernel in an ISA listing.  usage: isa_mix.py file.s kernel-name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] not in m.group(1): continue
    c = Counter(); reads = 0; groups = []
    for l in m.group(2).split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;': continue
        op = t.split()[0]
        if op.startswith('v_accvgpr'): c['accmov'] += 1
        elif op.startswith('scratch'): c['scratch'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
            if 'read' in op or 'permute' in op: reads += 1
        elif op.startswith('s_waitcnt'):
            c['wait'] += 1
            if 'lgkmcnt' in t and reads: groups.append(reads); reads = 0
        elif 'dpp' in t: c['dpp'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1; c['f64'] += 'f64' in op
        elif op.startswith('s_'): c['salu'] += 1
        else: c['other'] += 1
    print(m.group(1)[:60], sum(v for k, v in c.items() if k != 'f64'), dict(c))
    if groups: print('  lds reads per lgkm wait: n=%d mean=%.2f' % (len(groups), sum(groups) / len(groups)))
