"""Static audit of exposed memory waits in a kernel's ISA: every s_waitcnt with the distance (instructions) back to the
youngest load of the class it waits for.  usage: wait_audit.py file.s kernel_substring [max_distance]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
maxd = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lines = open(src).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l and ':' in l)
body = []
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
    else:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
        if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
loc = ''
for l in lines[i0 + 1:]:
    t = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m: loc = '%s:%s' % (files.get(int(m.group(1)), m.group(1)), m.group(2)); continue
    if t.startswith('s_endpgm'): body.append(t); break
    if not t or t.startswith(';') or t.startswith('.'): 
        if t.startswith('.LBB') or re.match(r'^\.?L?BB\d+_\d+:', t): body.append(t)
        continue
    body.append(t + '   ; ' + loc)
last = {'vm': None, 'lgkm_s': None, 'lgkm_l': None}
n = 0; tight = []
for t in body:
    if t.endswith(':'):
        last = {k: None for k in last}  # unknown across labels (conservative: forget)
        continue
    n += 1
    op = t.split()[0]
    if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'): last['vm'] = (n, t)
    elif op.startswith('s_load') or op.startswith('s_buffer_load'): last['lgkm_s'] = (n, t)
    elif op.startswith('ds_read') or op.startswith('ds_bpermute') or op.startswith('ds_swizzle') or (op.startswith('ds_') and 'rtn' in op): last['lgkm_l'] = (n, t)
    elif op == 's_waitcnt':
        for cls, key in (('vmcnt', ['vm']), ('lgkmcnt', ['lgkm_s', 'lgkm_l'])):
            if cls in t:
                for k in key:
                    if last[k] is not None and n - last[k][0] <= maxd:
                        tight.append((n, k, n - last[k][0], last[k][1]))
                        last[k] = None
print('instructions (static):', n)
from collections import Counter
c = Counter(k for _, k, _, _ in tight)
print('tight waits (<= %d instructions after the load):' % maxd, dict(c))
for n_, k, dist, ld in tight:
    print('%6d %-7s +%d  %s' % (n_, k, dist, ld[:120]))
