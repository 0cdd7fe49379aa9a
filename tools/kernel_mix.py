import re, sys
from collections import Counter
s=open('/root/repo/build/lld_ba.s').read()
for name in ['ba_linearize_pt_kernel','ba_linearize_ln_kernel','ba_backsub_pt_kernel','ba_backsub_ln_kernel']:
    i=s.index('\n_ZN5lldba%d%s'%(len(name),name)+'E')
    i=s.index(':\n',i)
    j=s.index('s_endpgm',i)
    body=s[i:j]
    c=Counter()
    for line in body.split('\n'):
        t=line.strip().split()
        if not t or t[0].startswith(('.',';')) or t[0].endswith(':'): continue
        op=t[0]
        if op.startswith('ds_'): c[op]+=1
        elif op.startswith('v_') and 'f64' in op: c['v_*f64']+=1
        elif 'dpp' in line: c['dpp']+=1
        elif op.startswith(('global_','buffer_','flat_')): c['_'.join(op.split('_')[:2])]+=1
        elif op.startswith('v_'): c['v_other']+=1
        elif op.startswith('s_waitcnt'): c['s_waitcnt']+=1
        elif op.startswith('s_'): c['s_other']+=1
    print(name, sum(c.values()), dict(c))
