"""The register-renaming pass of the build (famsa_amd/csrc/recolor_vgprs.py) must not change what a kernel
computes.  On the GPU the parity suite says so for every instantiation; here, without a GPU, the big straight-line
loop bodies of a few instantiations are EXECUTED for one lane -- a small interpreter of the dozen instructions they
consist of -- once as the compiler wrote them and once after the pass, from corresponding random register states;
every register the compiler's code names must end with the same content (at the place the renaming gave it)."""
import json
import os
import random
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "famsa_amd", "csrc")
DEV_S = os.path.join(CSRC, "_obj", "lcs_kernels.dev.s")
sys.path.insert(0, CSRC)

M32 = 0xFFFFFFFF
KERNELS = ["_ZN6lcsgpu20lcs_rows_kernel_pipeILi13ELi4ELi4EEEvNS_8RowsArgsE",   # the bench kernel (400 aa)
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi4ELi4ELi4EEEvNS_8RowsArgsE",    # short refs: partial last chunk
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi25ELi2ELi4EEEvNS_8RowsArgsE",
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi40ELi1ELi4EEEvNS_8RowsArgsE",
           "_ZN6lcsgpu15lcs_long_kernelILb0EEEvNS_8RowsArgsEPti"]


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
    import recolor_vgprs as R
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


def _all_kernels():
    """every kernel the build renames (Makefile: --only 'lcs_rows_kernel_pipe|lcs_long_kernel')"""
    if not os.path.exists(DEV_S):
        return list(KERNELS)
    names = re.findall(r"^(_ZN6lcsgpu\w*(?:lcs_rows_kernel_pipe|lcs_long_kernel)\w*):", open(DEV_S).read(), re.M)
    return sorted(set(names) | set(KERNELS))


ALL_KERNELS = _all_kernels()


@pytest.fixture(scope="module")
def recolored(tmp_path_factory):
    if not os.path.exists(DEV_S):
        pytest.skip("famsa_amd/csrc/_obj/lcs_kernels.dev.s is made by the build (make -C famsa_amd/csrc)")
    d = tmp_path_factory.mktemp("recolor")
    out, mp = str(d / "rec.s"), str(d / "map.json")
    only = "|".join(re.escape(k) for k in ALL_KERNELS)
    subprocess.check_call([sys.executable, os.path.join(CSRC, "recolor_vgprs.py"), DEV_S, out, "--only", only, "--map", mp],
                          stdout=subprocess.DEVNULL)
    return open(DEV_S).read().split("\n"), open(out).read().split("\n"), json.load(open(mp))


@pytest.mark.parametrize("kernel", ALL_KERNELS)
def test_loop_bodies_compute_the_same(recolored, kernel):
    old_all, new_all, maps = recolored
    if kernel not in maps:
        assert kernel not in KERNELS
        pytest.skip("no three-source instruction in this kernel: the pass leaves it alone")
    perm = {int(k): v for k, v in maps[kernel].items()}
    assert sorted(perm.values()) == sorted(perm)  # a permutation of the register file the kernel owns
    old_blocks, new_blocks = blocks_of(body(old_all, kernel)), blocks_of(body(new_all, kernel))
    assert len(old_blocks) == len(new_blocks)
    named = set()
    for ln in body(old_all, kernel):
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", ln.split(";")[0]):
            named.update([int(m.group(1))] if m.group(1) else range(int(m.group(2)), int(m.group(3)) + 1))
    checked = 0
    for ob, nb in zip(old_blocks, new_blocks):
        n3 = sum(1 for ln in ob if re.match(r"\s+v_bitop3_b32 v\d+, v\d+, v\d+, v\d+ ", ln))
        assert [l.split()[0] for l in ob if l.strip()] == [l.split()[0] for l in nb if l.strip()]  # same instructions, same order
        if n3 < 32:
            continue
        for trial in range(3 if kernel in KERNELS else 1):  # every instantiation once, the listed ones three times
            rng = random.Random(1000 * trial + n3)
            init = {r: rng.getrandbits(32) for r in perm}
            old_lane = Lane(init)
            new_lane = Lane({perm[r]: init[r] for r in init})
            run_block(ob, old_lane)
            run_block(nb, new_lane)
            bad = [r for r in sorted(named) if old_lane.v[r] != new_lane.v[perm[r]]]
            assert not bad, (kernel, n3, bad[:10])
            assert old_lane.vcc == new_lane.vcc
            assert old_lane.stores == new_lane.stores
        checked += 1
    assert checked >= 1


def test_pass_is_a_renaming_only(recolored):
    """Outside register operands nothing changes: same lines, same opcodes, same labels; of the descriptors only
    the VGPR counts."""
    old_all, new_all, _ = recolored
    assert len(old_all) == len(new_all)
    vtok = re.compile(r"\bv\d+\b|\bv\[\d+:\d+\]")
    other = set()
    for a, b in zip(old_all, new_all):
        if a == b:
            continue
        ca, cb = a.split(";")[0], b.split(";")[0]
        if vtok.sub("V", ca) == vtok.sub("V", cb):
            continue
        assert re.sub(r"\d+", "N", a) == re.sub(r"\d+", "N", b), (a, b)
        other.add(a.split()[0])
    assert other <= {".amdhsa_next_free_vgpr", ".amdhsa_accum_offset", ".set", ".vgpr_count:", ";"}, other
