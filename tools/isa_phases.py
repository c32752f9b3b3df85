"""dev tool: instruction mix between consecutive s_memtime reads inside one kernel (timing build)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] not in m.group(1): continue
    c = Counter(); seg = 0
    print(m.group(1)[:70])
    for l in m.group(2).split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;': continue
        op = t.split()[0]
        if op == 's_memtime':
            print('  seg %2d: total %5d %s' % (seg, sum(v for k, v in c.items() if k != 'f64'), dict(c)))
            c = Counter(); seg += 1; continue
        if op.startswith('v_accvgpr'): c['accmov'] += 1
        elif op.startswith('scratch'): c['scratch'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_nop'): c['nop'] += 1
        elif 'dpp' in t: c['dpp'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1; c['f64'] += 'f64' in op
        elif op.startswith('s_'): c['salu'] += 1
        else: c['other'] += 1
    print('  tail  : total %5d %s' % (sum(v for k, v in c.items() if k != 'f64'), dict(c)))
