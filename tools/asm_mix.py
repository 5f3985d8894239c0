"""Instruction mix of the largest basic block (the inner loop) of each kernel in a gfx950 .s file."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):\s*(?:;.*)?$', s, re.M):
    name = m.group(1)
    if pat not in name: continue
    a = m.end(); b = s.index('.Lfunc_end', a)
    body = s[a:b]
    blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
    best = None
    for i in range(1, len(blocks), 2):
        ins = [l.strip().split()[0] for l in blocks[i + 1].split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
        if best is None or len(ins) > len(best[1]): best = (blocks[i], ins)
    if best is None: continue
    c = collections.Counter(best[1])
    print(name[:90]); print('  ', best[0], len(best[1]), dict(c.most_common(20)))
