"""Memory instructions and waits of one kernel, in program order (reads a hipcc -S --cuda-device-only listing):
   python scripts/isa_mem.py /tmp/sp.s k_apply_sgdILi16ELb1"""
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):', txt, re.M)
name = m.group(1)
i = txt.index(name + ':'); j = txt.index('.Lfunc_end', i)
n = 0
for l in txt[i:j].split('\n'):
    t = l.strip()
    if t.startswith('.LBB'):
        print(t); continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    n += 1
    op = t.split()[0]
    if op.startswith(('global_', 'flat_', 'buffer_', 's_waitcnt', 's_barrier', 'ds_')) or 'cbranch' in op or op == 's_endpgm':
        print(n, t[:110])
print(n, 'instructions')
