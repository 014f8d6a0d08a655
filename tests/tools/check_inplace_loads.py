"""ISA check for k_write2 (tests/tools/isa.sh output): the texts requested ahead are loaded by inline asm into the registers the
lane's text already lives in, so the compiler does not know they are pending.  Walk the kernel's control-flow graph and make sure no
instruction reads one of those registers between such a load and the next `s_waitcnt vmcnt(0)` (or the counted wait of
texts_arrived_in_front_of, marked "texts arrived": vmcnt(N) behind N page stores issued after the loads).
usage: python tests/tools/check_inplace_loads.py /tmp/gdb_pipeline.s [kernel-name-substring]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else 'k_write2ILi1ELi8192E'
start = next(i for i, l in enumerate(src) if re.match(r'^_ZN.*' + want + r'.*:', l))
body = []
for l in src[start + 1:]:
    if l.startswith('.Lfunc_end'): break
    body.append(l)
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
# basic blocks
blocks, cur, name = {}, [], 'entry'
order = []
for l in body:
    marked = 'in-place text' in l
    counted = 'texts arrived' in l        # s_waitcnt vmcnt(N) behind N page stores issued after the loads (texts_arrived_in_front_of)
    t = l.split(';')[0].rstrip()
    m = re.match(r'^(\.LBB\w+):', t)
    if m:
        blocks[name] = cur; order.append(name); name = m.group(1); cur = []
        continue
    t = t.strip()
    if not t or t.startswith('.'): continue
    cur.append(t + (' ;INPLACE' if marked else '') + (' ;ARRIVED' if counted else ''))
blocks[name] = cur; order.append(name)
succ = {}
for i, b in enumerate(order):
    s = []
    ins = blocks[b]
    fall = True
    for t in ins:
        m = re.match(r's_cbranch_\w+\s+(\.LBB\w+)', t)
        if m: s.append(m.group(1))
        m = re.match(r's_branch\s+(\.LBB\w+)', t)
        if m: s.append(m.group(1)); fall = False
        if t.startswith('s_endpgm'): fall = False
    if fall and i + 1 < len(order): s.append(order[i + 1])
    succ[b] = s
# the asm loads: global_load_dwordx4 with an explicit "offset:" printed in decimal 0 / 16 / .. or hex (inline asm prints what was given)
inplace = re.compile(r'global_load_dwordx4 (v\[\d+:\d+\]), (v\[\d+:\d+\]), off offset:\w+ ;INPLACE$')
def sources(t):
    parts = t.replace(' ;INPLACE', '').split(None, 1)
    if len(parts) < 2: return set()
    op, args = parts
    ops = [x.strip() for x in args.split(',')]
    if op.startswith(('global_store', 'ds_write', 'scratch_store', 'buffer_store', 'flat_store')) or (op.startswith('v_cmp') and op.endswith('_e32')):
        srcs = ops
    else:
        srcs = ops[1:]
    out = set()
    for o in srcs: out |= regs(o)
    return out
# which registers are text registers: destinations of the in-place loads
txt = set()
for b in order:
    for t in blocks[b]:
        m = inplace.match(t)
        if m: txt |= regs(m.group(1))
print('in-place text registers:', sorted(txt))
state_in = {b: None for b in order}   # frozenset of pending regs at block entry
state_in['entry'] = frozenset()
work = ['entry']
bad = []
while work:
    b = work.pop()
    pend = set(state_in[b])
    for t in blocks[b]:
        if t.startswith('s_waitcnt') and ('vmcnt(0)' in t or ';ARRIVED' in t): pend.clear(); continue
        m = inplace.match(t)
        rd = sources(t)
        if rd & pend: bad.append((b, t, sorted(rd & pend)))
        if m: pend |= regs(m.group(1))
    for s in succ[b]:
        new = frozenset(pend) | (state_in[s] or frozenset())
        if state_in[s] is None or new != state_in[s]:
            state_in[s] = new; work.append(s)
seen = set()
for b, t, r in bad:
    if (b, t) in seen: continue
    seen.add((b, t)); print('READ OF A PENDING TEXT REGISTER in', b, ':', t, r)
print('violations:', len(seen))
sys.exit(1 if seen else 0)
