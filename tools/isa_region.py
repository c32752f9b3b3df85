"""Instruction mix between the two s_memtime reads of every kernel in an ISA listing (development aid)."""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
kern = None; inside = False; c = None
for l in lines:
    mk = re.match(r'^(_Z\S+):', l)
    if mk: kern = mk.group(1); seen = 0; continue
    t = l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;': continue
    op = t.split()[0]
    if op == 's_memtime':
        seen += 1
        if seen == 1: inside = True; c = Counter(); reads_since_wait = 0; waits_with_reads = 0
        else:
            inside = False
            tot = sum(v for k, v in c.items() if k not in ('f64',))
            print(kern[:24], 'total', tot, dict(c))
        continue
    if not inside: continue
    if op.startswith('v_accvgpr'): c['accmov'] += 1
    elif op.startswith('v_'):
        c['valu'] += 1
        if 'f64' in op: c['f64'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith('s_waitcnt'): c['wait'] += 1
    elif op.startswith('s_nop'): c['nop'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith(('scratch', 'buffer', 'global', 'flat')): c['mem'] += 1
    else: c['other'] += 1
