#!/usr/bin/env python3
"""recolor_vgprs.py -- give the three-source instructions of the hot LCS kernels sources in three different
VGPR banks.  A post-pass over the compiler's own assembly (hipcc -S), run by the Makefile.

Why.  Measured on gfx950 (scripts/ubench_banks.hip, 4 waves per SIMD, cycles per wave-instruction): a VALU
instruction with three VGPR sources issues in 2.6 cycles when the sources sit in three different register
banks (bank = register number mod 4) and in 4.3 when any two share a bank; two-source instructions (2.2) do
not care.  One 32-bit half-word step of lcs_rows_kernel_pipe is
    v_and_b32 t, V, M ; v_addc_co_u32 t, vcc, V, t, vcc ; v_bitop3_b32 V, t, V, M        (X = V2 | (V & ~M))
and hipcc's allocator, which knows nothing of banks, left 70 % of those v_bitop3_b32 with a shared bank
(583 of 832 in the 400-residue instantiation): 8.5 instead of 7.1 cycles per step.

What.  Registers are only names.  Two renamings that cannot change what a kernel computes:
  1. LOCAL: inside one straight-line block, a value that is defined there and dead before the block ends
     (the temporaries t, most of the mask words M loaded from LDS) may live in ANY register that is free
     between its definition and its last use.  The pass recomputes these live ranges from the text, takes
     the values out and puts them back one by one in definition order, each into the free register whose
     bank suits the three-source instructions that read it (even-aligned pairs stay even-aligned pairs).
  2. GLOBAL: a permutation of register numbers through the whole function (v0, where the work-item id
     arrives, and every register tuple's order and alignment class kept), searched greedily for what is left
     (the loop-carried X registers against the mask registers that cross the loop's back edge).
The register count and everything else in the kernel descriptor stay what they were.  Exactness is not a
matter of trust: the GPU parity suite compares every instantiation with the oracle bit for bit.

usage: recolor_vgprs.py in.s out.s [--only REGEX] [--report] [--map map.json]
"""
import random
import re
import sys

VTOK = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
KERNEL_LABEL = re.compile(r"^(_ZN6lcsgpu\w+):\s*(;.*)?$")
NO_VGPR_DEF = ("v_cmp", "v_cmpx", "ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "v_readlane",
               "v_readfirstlane", "s_", "global_atomic", "ds_add", "ds_min", "ds_max", "v_nop", "buffer_wbl2", "buffer_inv")
BLOCK_END = re.compile(r"\s+(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)")
# The LOCAL pass only touches blocks made of instructions it understands completely -- one destination that is the
# first operand and is written whole, every other VGPR token a plain source, no static hazard that depends on WHICH
# register an operand is (none of these forms has one on gfx950: the LDS / global results are guarded by counters,
# the carries by the order of the instructions, and neither changes) -- which is exactly the instruction set of the
# build's equivalence check (recolor_check.py executes such a block before and after the pass).  Anything else
# (tied or partial destinations: v_fmac / v_mac / v_dot*, *_d16[_hi] loads, v_permlane*_swap, DPP, op_sel or an SDWA
# dst_sel, trans ops, MFMA, inline assembly with text) leaves the block as the compiler wrote it.
ALLOWED_VALU = {
    "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_add_co_u32_e32", "v_addc_co_u32_e32", "v_add_co_u32_e64",
    "v_addc_co_u32_e64", "v_bitop3_b32", "v_add_u32_sdwa", "v_mov_b32_e32", "v_add_u32_e32", "v_sub_u32_e32",
    "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_lshl_or_b32", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_u32",
    "v_lshl_add_u64", "v_cndmask_b32_e32",
}
ALLOWED_MEM = {"ds_read_b64", "ds_read_b32", "global_load_ushort", "global_load_dword", "global_load_dwordx4",
               "global_store_short", "global_store_dword"}
ALLOWED_CMP = re.compile(r"v_cmp_(eq|ne|lt|gt|le|ge)_[ui]32_e32$")
# The GLOBAL pass renames registers through a whole function: that is a bijection of names and safe for every
# instruction that names its registers in its text -- not for relative addressing or the accumulation file
NO_PERMUTE = ("v_movrel", "s_set_gpr_idx", "v_accvgpr", "v_mfma", "s_setreg", "scratch_", "buffer_load", "buffer_store")


def understood(op, line):
    if op.startswith("s_"):
        return True
    if op == "v_add_u32_sdwa":
        return "dst_sel:DWORD" in line and "src0_sel:DWORD" in line and "dst_unused" in line and "UNUSED_PAD" in line
    return op in ALLOWED_VALU or op in ALLOWED_MEM or bool(ALLOWED_CMP.match(op))
LABEL = re.compile(r"^[.\w$]+:")


def split_code_comment(line):
    i = line.find(";")
    return (line, "") if i < 0 else (line[:i], line[i:])


def banks_differ(regs):
    return len({r % 4 for r in regs}) == len(regs)


class Operand:
    __slots__ = ("start", "end", "base", "width", "is_def")

    def __init__(self, start, end, base, width, is_def):
        self.start, self.end, self.base, self.width, self.is_def = start, end, base, width, is_def


def parse_instr(code):
    """-> (opcode, [Operand]) or None for non-instructions."""
    stripped = code.strip()
    if not stripped or stripped.startswith(".") or LABEL.match(stripped):
        return None
    opcode = stripped.split()[0]
    first_comma = code.find(",")
    has_def = not opcode.startswith(NO_VGPR_DEF)
    ops = []
    for m in VTOK.finditer(code):
        if m.group(1) is not None:
            base, width = int(m.group(1)), 1
        else:
            base, width = int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        is_def = has_def and (first_comma < 0 or m.start() < first_comma) and m.start() > code.find(opcode)
        ops.append(Operand(m.start(), m.end(), base, width, is_def))
    return opcode, ops


def relocate_block(lines, nv, stats, nv_used=None):
    """Local pass over one straight-line block.  nv_used: the set of registers the function names anywhere (a register
    outside it holds nothing).  Returns the new lines (or the old ones if anything unusual is seen)."""
    instrs = []  # (line index, opcode, operands)
    for li, ln in enumerate(lines):
        code, _ = split_code_comment(ln)
        p = parse_instr(code)
        if p:
            instrs.append((li, p[0], p[1]))
    n3 = sum(1 for _, op, ops in instrs if op == "v_bitop3_b32" and sum(1 for o in ops if not o.is_def) == 3)
    if n3 < 32:
        return lines
    # the region in which exec is known to stay put: between the last exec write near the top and the first later one
    exec_writes = [k for k, (li, op, ops) in enumerate(instrs)
                   if re.search(r"\bexec\b", split_code_comment(lines[li])[0].split(",")[0]) and op.startswith("s_")
                   or op.startswith(("v_cmpx", "s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec"))]
    lo = 0
    hi = len(instrs)
    for k in exec_writes:
        if k < len(instrs) // 2:
            lo = max(lo, k + 1)
        else:
            hi = min(hi, k)
    for li, op, _ in instrs:
        if not understood(op, lines[li]) or "op_sel" in lines[li] or "dpp" in lines[li]:
            stats["skipped_blocks"] = stats.get("skipped_blocks", 0) + 1
            stats.setdefault("skipped_ops", set()).add(op)
            return lines  # an instruction outside the allow-list: leave the block as the compiler wrote it

    # value numbering over the whole block
    class Val:
        __slots__ = ("vid", "base", "width", "d", "u", "uses", "fixed", "new")

    vals = []

    def new_val(base, width, d):
        v = Val()
        v.vid, v.base, v.width, v.d, v.u, v.uses, v.fixed, v.new = len(vals), base, width, d, d, [], False, base
        vals.append(v)
        return v

    cur = {}  # reg -> (Val, offset)
    for r in range(nv + 8):
        v = new_val(r, 1, -1)
        v.fixed = True
        cur[r] = (v, 0)
    op_val = {}  # (instr k, operand index) -> (Val, offset)
    for k, (li, op, ops) in enumerate(instrs):
        for oi, o in enumerate(ops):
            if o.is_def:
                continue
            parts = [cur[o.base + t] for t in range(o.width)]
            v0, off0 = parts[0]
            whole = all(p[0] is v0 and p[1] == off0 + t for t, p in enumerate(parts))
            if not whole:
                for p in parts:
                    p[0].fixed = True
            for p in parts:
                p[0].u = max(p[0].u, k)
            v0.uses.append((k, oi))
            op_val[(k, oi)] = (v0, off0) if whole else None
        for oi, o in enumerate(ops):
            if not o.is_def:
                continue
            v = new_val(o.base, o.width, k)
            if k < lo or k >= hi or o.width > 2:
                v.fixed = True
            for t in range(o.width):
                cur[o.base + t] = (v, t)
            op_val[(k, oi)] = (v, 0)
    end = len(instrs)
    for r, (v, _) in cur.items():  # whatever sits in a register at the end may be live after the block
        if v.d < 0 and nv_used is not None and r not in nv_used:
            v.u = -1  # a register the function never names: nothing lives there
            continue
        v.fixed = True
        v.u = end
    for v in vals:
        if not v.fixed and (v.u >= hi or v.d < lo):
            v.fixed = True

    # occupancy of the fixed values: register -> list of (d, u]
    occ = {r: [] for r in range(nv)}
    for v in vals:
        if v.fixed and not (v.d < 0 and v.u < 0):
            for t in range(v.width):
                if v.base + t < nv:
                    occ[v.base + t].append((v.d, v.u))
    movable = [v for v in vals if not v.fixed]
    placed = {v.vid for v in vals if v.fixed}
    def free(reg, d, u, me=None):
        u = max(u, d + 0.5)  # a definition occupies its register even if nothing reads it
        for a, b in occ[reg]:
            if a < u and d < max(b, a + 0.5):  # (a, b] and (d, u] overlap
                return False
        return True

    # three-source instructions: operand triples as (Val, offset)
    tri_of_val = {}
    tris = []
    for k, (li, op, ops) in enumerate(instrs):
        if op != "v_bitop3_b32":
            continue
        srcs = [op_val.get((k, oi)) for oi, o in enumerate(ops) if not o.is_def]
        if len(srcs) == 3 and all(s is not None for s in srcs):
            tris.append(srcs)
            for s in srcs:
                tri_of_val.setdefault(s[0].vid, []).append(len(tris) - 1)

    def reg_of(s):
        return s[0].new + s[1]

    def conflicts(v, cand):
        c = 0
        for ti in tri_of_val.get(v.vid, ()):
            regs = []
            for s in tris[ti]:
                if s[0] is v:
                    regs.append(cand + s[1])
                elif s[0].vid in placed:
                    regs.append(reg_of(s))
            if not banks_differ(regs):
                c += 1
        return c

    before = sum(1 for t in tris if not banks_differ([s[0].base + s[1] for s in t]))
    for v in sorted(movable, key=lambda x: x.d):
        best, best_c = None, None
        cands = range(1, nv - v.width + 1) if v.width == 1 else range(2 - (v.base % 2 == 1), nv - 1, 2)
        if v.width == 2 and v.base % 2 == 1:
            cands = range(1, nv - 1, 2)
        for cand in cands:
            if not all(free(cand + t, v.d, v.u, v) for t in range(v.width)):
                continue
            c = conflicts(v, cand)
            # a single goes, bank permitting, where it does not break up a free even pair (the mask words need those)
            hole = 0 if v.width == 2 or (cand ^ 1) >= nv or not free(cand ^ 1, v.d, v.u, v) else 1
            key = (c, hole, 0 if cand == v.base else 1, cand)
            if best is None or key < best_c:
                best, best_c = cand, key
                if c == 0 and cand == v.base:
                    break
        if best is None:
            return lines  # cannot happen when starting from a valid allocation with spare registers; keep the original
        v.new = best
        placed.add(v.vid)
        for t in range(v.width):
            occ[best + t].append((v.d, v.u))
    after = sum(1 for t in tris if not banks_differ([reg_of(s) for s in t]))
    stats["local"] = stats.get("local", 0) + (before - after)

    # rewrite
    out = list(lines)
    for k, (li, op, ops) in enumerate(instrs):
        code, comment = split_code_comment(lines[li])
        pieces, pos = [], 0
        for oi, o in enumerate(ops):
            ov = op_val.get((k, oi))
            pieces.append(code[pos:o.start])
            if ov is None or ov[0].new == ov[0].base:
                pieces.append(code[o.start:o.end])
            else:
                nb = ov[0].new + ov[1]
                pieces.append("v%d" % nb if o.width == 1 else "v[%d:%d]" % (nb, nb + o.width - 1))
            pos = o.end
        pieces.append(code[pos:])
        out[li] = "".join(pieces) + comment
    return out


def permute_function(lines, name, rng):
    used, tuples, tri = set(), [], []
    for ln in lines:
        code, _ = split_code_comment(ln)
        if code.lstrip().startswith("."):
            continue
        for m in VTOK.finditer(code):
            if m.group(1) is not None:
                used.add(int(m.group(1)))
            else:
                a, b = int(m.group(2)), int(m.group(3))
                used.update(range(a, b + 1))
                tuples.append((a, b))
        m = re.match(r"\s+v_bitop3_b32 v\d+, (v\d+), (v\d+), (v\d+) bitop3", code)
        if m:
            tri.append(tuple(int(x[1:]) for x in m.groups()))
    if not tri:
        return lines, None
    nv = max(used) + 1
    parent = list(range(nv))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in tuples:
        for r in range(a + 1, b + 1):
            parent[find(r)] = find(a)
    groups = {}
    for r in range(nv):
        groups.setdefault(find(r), []).append(r)
    blocks = []
    for g in groups.values():
        g.sort()
        assert g == list(range(g[0], g[0] + len(g))), f"{name}: tuple union not contiguous: {g}"
        blocks.append((g[0], len(g)))
    blocks.sort()
    cls = {}
    for base, size in blocks:
        if base != 0:
            cls.setdefault((size, base % 4 if size > 1 else -1), []).append(base)
    new_base = {base: base for base, _ in blocks}
    block_of = {}
    for base, size in blocks:
        for r in range(base, base + size):
            block_of[r] = base

    def reg_now(r):
        b = block_of[r]
        return new_base[b] + (r - b)

    touch = {}
    for i, t in enumerate(tri):
        for r in t:
            touch.setdefault(block_of[r], set()).add(i)

    def cost_of(idx):
        return sum(1 for i in idx if not banks_differ([reg_now(r) for r in tri[i]]))

    start = cost_of(range(len(tri)))
    best = start
    improved, rounds = True, 0
    while improved and best > 0 and rounds < 40:
        improved, rounds = False, rounds + 1
        for (size, _), members in cls.items():
            if len(members) < 2:
                continue
            order = [m for m in members if m in touch]
            rng.shuffle(order)
            for a in order:
                for b in members:
                    if a == b:
                        continue
                    idx = touch.get(a, set()) | touch.get(b, set())
                    before = cost_of(idx)
                    if before == 0:
                        break
                    new_base[a], new_base[b] = new_base[b], new_base[a]
                    after = cost_of(idx)
                    if after < before:
                        best += after - before
                        improved = True
                    else:
                        new_base[a], new_base[b] = new_base[b], new_base[a]
    mapping = {r: reg_now(r) for r in range(nv)}
    assert sorted(mapping.values()) == list(range(nv)) and mapping[0] == 0

    def sub(m):
        if m.group(1) is not None:
            return "v%d" % mapping[int(m.group(1))]
        a, b = int(m.group(2)), int(m.group(3))
        assert mapping[b] - mapping[a] == b - a
        return "v[%d:%d]" % (mapping[a], mapping[b])

    out = []
    for ln in lines:
        code, comment = split_code_comment(ln)
        out.append(ln if code.lstrip().startswith(".") else VTOK.sub(sub, code) + comment)
    return out, (len(tri), start, best, nv, mapping)


def count_conflicts(lines):
    n = c = 0
    for ln in lines:
        m = re.match(r"\s+v_bitop3_b32 v\d+, v(\d+), v(\d+), v(\d+) bitop3", split_code_comment(ln)[0])
        if m:
            n += 1
            c += 0 if banks_differ([int(x) for x in m.groups()]) else 1
    return n, c


def recolor_function(lines, name, rng):
    n, c0 = count_conflicts(lines)
    if n == 0:
        return lines, None
    for ln in lines:
        code = split_code_comment(ln)[0].strip()
        if code and code.split()[0].startswith(NO_PERMUTE):
            print(f"recolor_vgprs: {name}: '{code.split()[0]}' found -- kernel left as compiled", file=sys.stderr)
            return lines, None
    nv0 = 0
    for ln in lines:
        for m in VTOK.finditer(split_code_comment(ln)[0]):
            nv0 = max(nv0, int(m.group(1)) + 1 if m.group(1) is not None else int(m.group(3)) + 1)
    # Registers are granted in blocks of 8 and a SIMD holds 512 per lane: the kernel may use every register up to the
    # largest multiple of 8 that allows as many waves per SIMD as its present count does -- free seats for the local pass.
    alloc = (nv0 + 7) // 8 * 8
    waves = min(8, 512 // alloc)
    nv = alloc
    while nv + 8 <= 256 and min(8, 512 // (nv + 8)) == waves:
        nv += 8
    def local_pass(src):
        # the registers the function names NOW (after an earlier turn's renamings the free seats above the compiler's
        # count may hold values that only pass through a block: those are not free there)
        named = set()
        for ln in src:
            for m in VTOK.finditer(split_code_comment(ln)[0]):
                if m.group(1) is not None:
                    named.add(int(m.group(1)))
                else:
                    named.update(range(int(m.group(2)), int(m.group(3)) + 1))
        out, blk, st = [], [], {}
        for ln in src:
            code = split_code_comment(ln)[0]
            if LABEL.match(code.strip()) and blk:
                out.extend(relocate_block(blk, nv, st, named))
                blk = []
            blk.append(ln)
            if BLOCK_END.match(code):
                out.extend(relocate_block(blk, nv, st, named))
                blk = []
        if blk:
            out.extend(relocate_block(blk, nv, st, named))
        return out

    # alternate the two renamings while the count falls: the permutation moves the loop-carried registers to banks
    # that leave the local pass room, the local pass then re-seats the temporaries around them
    best, best_c, trail = lines, c0, [c0]
    cur = lines
    perm = {r: r for r in range(nv)}  # compiler's register number -> present number, for values the local pass leaves alone
    best_perm = dict(perm)
    for _ in range(4):
        cur = local_pass(cur)
        trail.append(count_conflicts(cur)[1])
        if trail[-1] < best_c:
            best, best_c, best_perm = cur, trail[-1], dict(perm)
        cur, pst = permute_function(cur, name, rng)
        if pst:
            perm = {r: pst[4].get(v, v) for r, v in perm.items()}
        trail.append(count_conflicts(cur)[1])
        if trail[-1] < best_c:
            best, best_c, best_perm = cur, trail[-1], dict(perm)
        if len(trail) >= 5 and trail[-1] >= trail[-3]:
            break
    used = 0
    for ln in best:
        for m in VTOK.finditer(split_code_comment(ln)[0]):
            used = max(used, int(m.group(1)) + 1 if m.group(1) is not None else int(m.group(3)) + 1)
    return best, (n, c0, trail[1], best_c, nv0, used, best_perm)


def _job(job):
    name, body = job
    best = None
    for seed in (12345, 777, 4242):  # fixed seeds: the build is reproducible; the greedy search profits from a few tries
        out, st = recolor_function(body, name, random.Random(seed))
        if st is None:
            return out, st
        if best is None or st[3] < best[1][3]:
            best = (out, st)
        if st[3] * 50 <= st[0]:
            break
    return best


def main():
    argv = sys.argv[1:]
    only, report, files, map_file = None, False, [], None
    i = 0
    while i < len(argv):
        if argv[i] == "--only":
            only = re.compile(argv[i + 1])
            i += 2
        elif argv[i] == "--report":
            report = True
            i += 1
        elif argv[i] == "--map":
            map_file = argv[i + 1]
            i += 2
        else:
            files.append(argv[i])
            i += 1
    src, dst = files
    lines = open(src).read().split("\n")
    # cut the file into plain stretches and kernel bodies; the bodies are independent jobs
    pieces, i = [], 0  # ("text", [lines]) | ("kernel", name, [lines])
    plain = []
    while i < len(lines):
        m = KERNEL_LABEL.match(lines[i])
        if not m or (only and not only.search(m.group(1))):
            plain.append(lines[i])
            i += 1
            continue
        j = i
        while j < len(lines) and "s_endpgm" not in lines[j]:
            j += 1
        if plain:
            pieces.append(("text", plain))
            plain = []
        pieces.append(("kernel", m.group(1), lines[i:j + 1]))
        i = j + 1
    if plain:
        pieces.append(("text", plain))
    jobs = [(p[1], p[2]) for p in pieces if p[0] == "kernel"]
    import multiprocessing as mp
    import os
    workers = max(1, min(len(jobs), len(os.sched_getaffinity(0)), 16))
    if workers > 1:
        with mp.Pool(workers) as pool:
            results = pool.map(_job, jobs, chunksize=1)
    else:
        results = [_job(j) for j in jobs]
    out, stats, k = [], [], 0
    for p in pieces:
        if p[0] == "text":
            out.extend(p[1])
        else:
            body, st = results[k]
            k += 1
            out.extend(body)
            if st:
                stats.append((p[1], st))
    # the kernel descriptors and the metadata of the kernels whose register count grew
    grew = {name: st[5] for name, st in stats if st[5] > st[4]}
    if grew:
        cur_kernel, patched = None, []
        for ln in out:
            m = re.match(r"\s*\.amdhsa_kernel (\S+)", ln)
            if m:
                cur_kernel = m.group(1)
            if ".end_amdhsa_kernel" in ln:
                cur_kernel = None
            if cur_kernel in grew:
                nvk = grew[cur_kernel]
                if ".amdhsa_next_free_vgpr" in ln:
                    ln = re.sub(r"\d+", str(nvk), ln, count=1)
                elif ".amdhsa_accum_offset" in ln:
                    ln = re.sub(r"\d+", str((nvk + 3) // 4 * 4), ln, count=1)
            m = re.match(r"\s*\.set (\S+)\.num_vgpr, \d+", ln)
            if m and m.group(1) in grew:
                ln = re.sub(r"\d+$", str(grew[m.group(1)]), ln)
            patched.append(ln)
        out, cur_kernel = [], None
        for ln in patched:  # amdhsa.kernels metadata: .name comes before .vgpr_count within an entry
            m = re.match(r"\s*\.name:\s+(\S+)", ln)
            if m:
                cur_kernel = m.group(1)
            if cur_kernel in grew and re.match(r"\s*\.vgpr_count:", ln):
                ln = re.sub(r"\d+", str(grew[cur_kernel]), ln, count=1)
                cur_kernel = None
            out.append(ln)
    open(dst, "w").write("\n".join(out))
    if map_file:  # for tests: where the global renaming put each of the compiler's registers
        import json
        json.dump({name: {str(k): v for k, v in st[6].items()} for name, st in stats}, open(map_file, "w"))
    if report:
        t = [sum(s[k] for _, s in stats) for k in range(4)]
        print(f"recolor_vgprs: {len(stats)} kernels, {t[0]} three-source v_bitop3_b32; with a shared bank: {t[1]} as compiled "
              f"-> {t[2]} after the local pass -> {t[3]} after the permutation", file=sys.stderr)
        for name, (n, a, b, c, nv0, nv1, _) in stats:
            if "Li13ELi4E" in name or c * 20 > n:
                print(f"  {name}: {n} ops, shared bank {a} -> {b} -> {c} ({nv0} -> {nv1} VGPRs)", file=sys.stderr)


if __name__ == "__main__":
    main()
