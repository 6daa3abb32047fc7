#!/usr/bin/env python3
"""Dev tool: approximate VGPR liveness over one kernel of a hipcc -S listing; prints the live-register count at the
entry of every basic block and the peak inside it (defs are treated as full kills).
usage: vgpr_pressure.py file.s <kernel-name-regex>"""
import re, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "famsa_amd", "csrc"))
from recolor_vgprs import parse_instr, split_code_comment

src, pat = sys.argv[1], re.compile(sys.argv[2])
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN6lcsgpu\w+:", l) and pat.search(l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start + 1:end]
blocks, cur, name = [], [], "entry"
labels = {}
for ln, l in enumerate(body):
    code, _ = split_code_comment(l)
    m = re.match(r"^(\.LBB\w+):", code.strip())
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    p = parse_instr(code)
    if p is None:
        continue
    cur.append((ln + start + 2, p, code.strip()))
    if re.match(r"\s*(s_cbranch|s_branch|s_endpgm)", code):
        blocks.append((name, cur))
        name, cur = f"{name}+{ln}", []
if cur:
    blocks.append((name, cur))
idx = {n: i for i, (n, _) in enumerate(blocks)}
succ = []
for i, (n, ins) in enumerate(blocks):
    s = []
    if ins:
        last = ins[-1][2]
        if last.startswith("s_branch"):
            s = [idx[last.split()[1]]]
        elif last.startswith("s_cbranch"):
            s = [idx[last.split()[1]]] + ([i + 1] if i + 1 < len(blocks) else [])
        elif last.startswith("s_endpgm"):
            s = []
        else:
            s = [i + 1] if i + 1 < len(blocks) else []
    else:
        s = [i + 1] if i + 1 < len(blocks) else []
    succ.append(s)
def regs(op):
    return range(op.base, op.base + op.width)
use_b, def_b = [], []
for n, ins in blocks:
    u, d = set(), set()
    for _, (opc, ops), _ in ins:
        for o in ops:
            if not o.is_def:
                u |= {r for r in regs(o) if r not in d}
        for o in ops:
            if o.is_def:
                d |= set(regs(o))
    use_b.append(u); def_b.append(d)
live_in = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for i in reversed(range(len(blocks))):
        out = set()
        for s in succ[i]:
            out |= live_in[s]
        li = use_b[i] | (out - def_b[i])
        if li != live_in[i]:
            live_in[i] = li; changed = True
for i, (n, ins) in enumerate(blocks):
    out = set()
    for s in succ[i]:
        out |= live_in[s]
    live = set(out); peak, at = len(live), None
    for ln, (opc, ops), code in reversed(ins):
        for o in ops:
            if o.is_def:
                live -= set(regs(o))
        for o in ops:
            if not o.is_def:
                live |= set(regs(o))
        if len(live) > peak:
            peak, at = len(live), ln
    if len(ins) > 20 or peak > 60:
        print(f"{n:24s} instrs {len(ins):5d} live_in {len(live_in[i]):3d} live_out {len(out):3d} peak {peak:3d} at line {at}")
if len(sys.argv) > 3:  # block name: list the registers that are live through it without being touched
    i = idx[sys.argv[3]]
    n, ins = blocks[i]
    touched = set()
    for _, (opc, ops), _ in ins:
        for o in ops:
            touched |= set(regs(o))
    thru = sorted(live_in[i] - touched)
    print("live-through untouched:", len(thru), thru)
    # where is each defined / used elsewhere
    for r in thru[:40]:
        sites = []
        for bn, bins in blocks:
            for ln, (opc, ops), code in bins:
                if any(r in regs(o) for o in ops):
                    sites.append((ln, code))
        print(f"v{r}:", "; ".join(f"{ln}:{c}" for ln, c in sites[:4]), "..." if len(sites) > 4 else "")
if len(sys.argv) > 4:  # register number: blocks where it is read before being written (upward exposed), and its successors chain
    r = int(sys.argv[4])
    for i, (n, ins) in enumerate(blocks):
        if r in use_b[i]:
            ln = next(l for l, (opc, ops), c in ins if any((not o.is_def) and r in regs(o) for o in ops))
            print("upward-exposed use of v%d in block %s (line %d), live_in there: %s" % (r, n, ln, r in live_in[i]))
