#!/usr/bin/env python3
"""recolor_check.py -- the build's equivalence check of the register-renaming pass (recolor_vgprs.py).

The pass rewrites the compiler's assembly of the hot LCS kernels; this check EXECUTES, for one lane, every
straight-line block the pass may have touched (the blocks with >= 32 three-source v_bitop3_b32) -- a small
interpreter of the dozen instructions such a block consists of -- once as the compiler wrote it and once after
the pass, from corresponding random register states, and requires every register the compiler's code names to end
with the same content at the place the renaming gave it, the same carry, and the same global stores in the same
order.  Outside register operands the two listings must be the same text.  An instruction the interpreter does
not know, in a block the pass changed, is a FAILURE (the pass's allow-list is this interpreter's instruction set):
the Makefile then builds the kernels as compiled and lcsgpu_version() reports "recolor=failed".

usage: recolor_check.py dev.s rec.s map.json      (exit status 0 = equivalent)
Also imported by tests/test_recolor.py.
"""
import json
import os
import random
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import recolor_vgprs as R  # noqa: E402

M32 = 0xFFFFFFFF


def mix(x):
    x = (x * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


class Lane:
    """One lane's registers.  SGPRs / literals are constants of the run; exec is ignored (one active lane)."""

    def __init__(self, vgpr):
        self.v = dict(vgpr)
        self.vcc = 0
        self.carry = {}  # SGPR pairs written by VALU carries / compares
        self.stores = []  # (address, data) of global stores, in program order

    def val(self, tok):
        tok = tok.strip()
        if re.fullmatch(r"v\d+", tok):
            return self.v[int(tok[1:])]
        if re.fullmatch(r"s\d+", tok):
            return mix(1000 + int(tok[1:])) & M32
        if tok == "vcc":
            return self.vcc
        if re.fullmatch(r"-?\d+", tok):
            return int(tok) & M32
        if re.fullmatch(r"0x[0-9a-fA-F]+", tok):
            return int(tok, 16) & M32
        raise ValueError(f"operand {tok!r}")

    def val64(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return self.v[int(m.group(1))] | (self.v[int(m.group(2))] << 32)
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            return mix(5000 + int(m.group(1)))
        return self.val(tok)

    def set(self, tok, x):
        self.v[int(tok.strip()[1:])] = x & M32

    def set64(self, tok, x):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok.strip())
        self.v[int(m.group(1))], self.v[int(m.group(2))] = x & M32, (x >> 32) & M32


def run_block(lines, lane):
    for ln in lines:
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        op, _, rest = code.partition(" ")
        if op.startswith("s_"):
            continue  # scalar code and waits: the same text in both versions, no VGPR involved
        mods = ""
        ops = [o.strip() for o in rest.split(",")]
        if " " in ops[-1]:
            ops[-1], _, mods = ops[-1].partition(" ")
        v = lane.val
        if op == "v_and_b32_e32":
            lane.set(ops[0], v(ops[1]) & v(ops[2]))
        elif op == "v_or_b32_e32":
            lane.set(ops[0], v(ops[1]) | v(ops[2]))
        elif op == "v_xor_b32_e32":
            lane.set(ops[0], v(ops[1]) ^ v(ops[2]))
        elif op in ("v_add_co_u32_e32", "v_addc_co_u32_e32"):
            assert ops[1] == "vcc"
            t = v(ops[2]) + v(ops[3]) + (lane.vcc if op.startswith("v_addc") else 0)
            lane.set(ops[0], t)
            lane.vcc = t >> 32
        elif op in ("v_add_co_u32_e64", "v_addc_co_u32_e64"):
            cin = 0
            if op.startswith("v_addc"):
                cin = lane.vcc if ops[4] == "vcc" else lane.carry.get(ops[4], mix(len(ops[4])) & 1)
            t = v(ops[2]) + v(ops[3]) + cin
            lane.set(ops[0], t)
            if ops[1] == "vcc":
                lane.vcc = t >> 32
            else:
                lane.carry[ops[1]] = t >> 32
        elif op == "v_bitop3_b32":
            a, b, c = v(ops[1]), v(ops[2]), v(ops[3])
            table = int(re.search(r"bitop3:(0x[0-9a-fA-F]+|\d+)", mods).group(1), 0)
            r = 0
            for i in range(32):
                idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1)
                r |= ((table >> idx) & 1) << i
            lane.set(ops[0], r)
        elif op in ("ds_read_b64", "ds_read_b32", "ds_read2_b32", "ds_read2_b64", "ds_read_b128"):
            assert op in ("ds_read_b64", "ds_read_b32"), code
            off = int(re.search(r"offset:(\d+)", mods).group(1)) if "offset:" in mods else 0
            x = mix((v(ops[1]) + off) & M32)
            (lane.set64 if op == "ds_read_b64" else lane.set)(ops[0], x)
        elif op in ("global_load_ushort", "global_load_dword", "global_load_dwordx4") and ops[2] == "off":
            # memory is a function of the address (the long-ref kernel's carry stream / residue chunks)
            off = int(re.search(r"offset:(-?\d+)", mods).group(1)) if "offset:" in mods else 0
            x = mix((lane.val64(ops[1]) + off) & 0xFFFFFFFFFFFFFFFF)
            if op == "global_load_dwordx4":
                m = re.fullmatch(r"v\[(\d+):(\d+)\]", ops[0])
                for i, r in enumerate(range(int(m.group(1)), int(m.group(2)) + 1)):
                    lane.v[r] = mix(x + i) & M32
            else:
                lane.set(ops[0], x & (0xFFFF if op.endswith("ushort") else M32))
        elif op in ("global_store_short", "global_store_dword") and ops[2] == "off":
            off = int(re.search(r"offset:(-?\d+)", mods).group(1)) if "offset:" in mods else 0
            lane.stores.append(((lane.val64(ops[0]) + off) & 0xFFFFFFFFFFFFFFFF,
                                v(ops[1]) & (0xFFFF if op.endswith("short") else M32)))
        elif op == "v_add_u32_sdwa":
            sel = re.search(r"src1_sel:BYTE_(\d)", mods)
            assert sel and "src0_sel:DWORD" in mods and "dst_sel:DWORD" in mods, code
            lane.set(ops[0], v(ops[1]) + ((v(ops[2]) >> (8 * int(sel.group(1)))) & 0xFF))
        elif op == "v_mov_b32_e32":
            lane.set(ops[0], v(ops[1]))
        elif op == "v_add_u32_e32":
            lane.set(ops[0], v(ops[1]) + v(ops[2]))
        elif op == "v_sub_u32_e32":
            lane.set(ops[0], v(ops[1]) - v(ops[2]))
        elif op == "v_lshlrev_b32_e32":
            lane.set(ops[0], v(ops[2]) << (v(ops[1]) & 31))
        elif op == "v_lshrrev_b32_e32":
            lane.set(ops[0], v(ops[2]) >> (v(ops[1]) & 31))
        elif op == "v_lshl_or_b32":
            lane.set(ops[0], (v(ops[1]) << (v(ops[2]) & 31)) | v(ops[3]))
        elif op == "v_lshl_add_u32":
            lane.set(ops[0], (v(ops[1]) << (v(ops[2]) & 31)) + v(ops[3]))
        elif op == "v_and_or_b32":
            lane.set(ops[0], (v(ops[1]) & v(ops[2])) | v(ops[3]))
        elif op == "v_bfe_u32":
            lane.set(ops[0], (v(ops[1]) >> (v(ops[2]) & 31)) & ((1 << (v(ops[3]) & 31)) - 1))
        elif op == "v_lshl_add_u64":
            lane.set64(ops[0], (lane.val64(ops[1]) << (v(ops[2]) & 7)) + lane.val64(ops[3]))
        elif op == "v_cndmask_b32_e32":
            lane.set(ops[0], v(ops[2]) if lane.vcc else v(ops[1]))
        elif op.startswith("v_cmp_") and op.endswith("_e32"):
            a, b = v(ops[-2]), v(ops[-1])
            lane.vcc = int({"eq": a == b, "ne": a != b, "lt": a < b, "gt": a > b, "le": a <= b, "ge": a >= b}[op.split("_")[2]])
        else:
            raise AssertionError("instruction the interpreter does not know: " + code)


def blocks_of(lines):
    out, blk = [], []
    for ln in lines:
        code = R.split_code_comment(ln)[0]
        if R.LABEL.match(code.strip()) and blk:
            out.append(blk)
            blk = []
        blk.append(ln)
        if R.BLOCK_END.match(code):
            out.append(blk)
            blk = []
    if blk:
        out.append(blk)
    return out


def body(lines, name):
    i = next(k for k, l in enumerate(lines) if l.startswith(name + ":"))
    j = i
    while "s_endpgm" not in lines[j]:
        j += 1
    return lines[i:j + 1]


VTOK_ANY = re.compile(r"\bv\d+\b|\bv\[\d+:\d+\]")
DESCRIPTOR_LINES = {".amdhsa_next_free_vgpr", ".amdhsa_accum_offset", ".set", ".vgpr_count:", ";"}


def renaming_only(old_all, new_all):
    """Outside register operands nothing may change: same lines, same opcodes, same labels; of the descriptors
    only the VGPR counts.  Returns an error string or None."""
    if len(old_all) != len(new_all):
        return "line count changed: %d -> %d" % (len(old_all), len(new_all))
    for a, b in zip(old_all, new_all):
        if a == b:
            continue
        ca, cb = a.split(";")[0], b.split(";")[0]
        if VTOK_ANY.sub("V", ca) == VTOK_ANY.sub("V", cb):
            continue
        if re.sub(r"\d+", "N", a) != re.sub(r"\d+", "N", b) or (a.split() or [""])[0] not in DESCRIPTOR_LINES:
            return "line changed beyond register names: %r -> %r" % (a, b)
    return None


def check_kernel(old_all, new_all, kernel, perm, trials=1):
    """-> (blocks checked, error string or None) for one renamed kernel."""
    if sorted(perm.values()) != sorted(perm):
        return 0, "the register map is not a permutation"
    try:
        old_blocks, new_blocks = blocks_of(body(old_all, kernel)), blocks_of(body(new_all, kernel))
    except StopIteration:
        return 0, "kernel not found in the listing"
    if len(old_blocks) != len(new_blocks):
        return 0, "block structure changed"
    named = set()
    for ln in body(old_all, kernel):
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", ln.split(";")[0]):
            named.update([int(m.group(1))] if m.group(1) else range(int(m.group(2)), int(m.group(3)) + 1))
    checked = 0
    for ob, nb in zip(old_blocks, new_blocks):
        if [l.split()[0] for l in ob if l.strip()] != [l.split()[0] for l in nb if l.strip()]:
            return checked, "instructions or their order changed"
        if ob == nb:
            continue  # untouched by the local pass and the permutation is the identity here
        n3 = sum(1 for ln in ob if re.match(r"\s+v_bitop3_b32 v\d+, v\d+, v\d+, v\d+ ", ln))
        if n3 < 32:
            continue  # the local pass leaves such blocks alone; the permutation is a bijection of names
        for trial in range(trials):
            rng = random.Random(1000 * trial + n3)
            init = {r: rng.getrandbits(32) for r in perm}
            old_lane = Lane(init)
            new_lane = Lane({perm[r]: init[r] for r in init})
            try:
                run_block(ob, old_lane)
                run_block(nb, new_lane)
            except (AssertionError, ValueError, KeyError, AttributeError) as e:
                return checked, "block of %d three-source ops: %s" % (n3, e)
            bad = [r for r in sorted(named) if old_lane.v[r] != new_lane.v[perm[r]]]
            if bad:
                return checked, "block of %d three-source ops: registers differ: %s" % (n3, bad[:10])
            if old_lane.vcc != new_lane.vcc:
                return checked, "block of %d three-source ops: carry differs" % n3
            if old_lane.stores != new_lane.stores:
                return checked, "block of %d three-source ops: global stores differ" % n3
        checked += 1
    return checked, None


def _job(args):
    old_all, new_all, kernel, perm = args
    return kernel, check_kernel(old_all, new_all, kernel, perm)


def main():
    dev_s, rec_s, map_json = sys.argv[1:4]
    old_all, new_all = open(dev_s).read().split("\n"), open(rec_s).read().split("\n")
    maps = json.load(open(map_json))
    err = renaming_only(old_all, new_all)
    if err:
        print("recolor_check: FAILED: " + err, file=sys.stderr)
        return 1
    if not maps:
        print("recolor_check: FAILED: the pass renamed no kernel (listing format changed?)", file=sys.stderr)
        return 1
    # cut each kernel's body out once: the workers get small inputs
    jobs = []
    for kernel, mp in maps.items():
        try:
            jobs.append((body(old_all, kernel), body(new_all, kernel), kernel, {int(k): v for k, v in mp.items()}))
        except StopIteration:
            print("recolor_check: FAILED: %s not found" % kernel, file=sys.stderr)
            return 1
    import multiprocessing as mp_
    workers = max(1, min(len(jobs), len(os.sched_getaffinity(0)), 16))
    if workers > 1:
        with mp_.Pool(workers) as pool:
            results = pool.map(_job, jobs, chunksize=1)
    else:
        results = [_job(j) for j in jobs]
    blocks = 0
    for kernel, (checked, err) in results:
        if err:
            print("recolor_check: FAILED: %s: %s" % (kernel, err), file=sys.stderr)
            return 1
        blocks += checked
    if blocks == 0:
        print("recolor_check: FAILED: no renamed block was executed", file=sys.stderr)
        return 1
    print("recolor_check: %d kernels, %d renamed loop bodies executed before and after the pass: equivalent" %
          (len(results), blocks), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
