#!/usr/bin/env python
"""Build-time check for kernels that issue vector-memory loads from inline asm (csrc/corr_pyramid.hip).

hipcc does not know that the destination registers of such loads are in flight until the matching explicit
s_waitcnt vmcnt(N); if register pressure makes it spill or copy one of them before the wait, the kernel reads
garbage (seen once: a `scratch_store ... Folded Spill` of a tap register).  This script replays the ISA of every
kernel whose name matches, models the in-order vmcnt counter and fails if an instruction other than the
explicit waits touches a register with a load still outstanding.

usage: audit_asm_loads.py <file.s> <kernel-name-substring>
"""
import re
import sys


def regs(tok):
    o = set()
    for m in re.finditer(r'\bv(\d+)\b', tok):
        o.add(int(m.group(1)))
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        o.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return o


def used_regs(s):
    """registers an instruction reads/writes; packed-f32 ops with op_sel_hi:[0,..] read only the low register of
    their first source pair"""
    u = regs(s)
    m = re.match(r'\s*(v_pk_\w+)\s+(.*?)\s+op_sel_hi:\[0,', s)
    if m:
        ops = m.group(2).split(',')
        if len(ops) >= 2:
            pair = re.search(r'v\[(\d+):(\d+)\]', ops[1])
            if pair:
                hi = int(pair.group(2))
                others = regs(','.join(ops[:1] + ops[2:]))
                if hi not in others:
                    u.discard(hi)
    return u


def audit(lines):
    out, viol = [], []
    for i, l in enumerate(lines):
        s = l.strip()
        op = s.split()[0]
        pend = set().union(*[d for _, d in out]) if out else set()
        if op.startswith(('global_load', 'scratch_load', 'buffer_load')):
            ops = s.split(None, 1)[1].split(',')
            d, a = regs(ops[0]), regs(','.join(ops[1:]))
            if a & pend:
                viol.append((i, s, sorted(a & pend)))
            if d & pend:
                viol.append((i, s, sorted(d & pend)))
            out.append((i, d))
            continue
        if op.startswith(('global_store', 'scratch_store', 'buffer_store')):
            u = regs(s)
            if u & pend:
                viol.append((i, s, sorted(u & pend)))
            out.append((i, set()))
            continue
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1))
                out = [] if n == 0 else (out[len(out) - n:] if len(out) > n else out)
            continue
        u = used_regs(s)
        if u & pend:
            viol.append((i, s, sorted(u & pend)))
    return viol


def main(path, pattern):
    text = open(path).read().split('\n')
    kernels, cur, name = {}, None, None
    for l in text:
        m = re.match(r'^(\S+):\s*(;.*)?$', l)
        if m and pattern in m.group(1) and not m.group(1).startswith('.'):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            t = l.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                cur.append(l)
            if t.startswith('s_endpgm'):
                kernels[name] = cur
                cur = None
    bad = 0
    for k, lines in kernels.items():
        v = audit(lines)
        print("%s: %d instructions, %d hazards" % (k, len(lines), len(v)))
        for i, s, r in v[:10]:
            print("   line %d: %s   [in flight: v%s]" % (i, s, r))
        bad += len(v)
    if not kernels:
        print("no kernel matching", pattern)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
