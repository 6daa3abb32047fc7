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
DEV_S_FUSED = os.path.join(CSRC, "_obj", "lcs_kernels_fused.dev.s")  # the second translation unit: FUSE instantiations
sys.path.insert(0, CSRC)

import recolor_check as RC  # the interpreter lives with the build: it is a build step (csrc/Makefile)

KERNELS = ["_ZN6lcsgpu20lcs_rows_kernel_pipeILi13ELi4ELi4ELb0EEEvNS_8RowsArgsE",   # the bench kernel (400 aa)
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi4ELi4ELi4ELb0EEEvNS_8RowsArgsE",    # short refs: partial last chunk
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi25ELi2ELi4ELb0EEEvNS_8RowsArgsE",
           "_ZN6lcsgpu20lcs_rows_kernel_pipeILi40ELi1ELi4ELb0EEEvNS_8RowsArgsE",
           "_ZN6lcsgpu15lcs_long_kernelILb0ELb0EEEvNS_8RowsArgsEPti"]


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
    checked, err = RC.check_kernel(old_all, new_all, kernel, perm, trials=3 if kernel in KERNELS else 1)
    assert err is None, (kernel, err)
    assert checked >= 1


def test_fused_unit_loop_bodies_compute_the_same(tmp_path):
    """The same check for the translation unit of the fused instantiations (the build runs it on all of them;
    here: the 13- and 4-half-word kernels and the long-ref kernel)."""
    if not os.path.exists(DEV_S_FUSED):
        pytest.skip("famsa_amd/csrc/_obj/lcs_kernels_fused.dev.s is made by the build (make -C famsa_amd/csrc)")
    out, mp = str(tmp_path / "rec.s"), str(tmp_path / "map.json")
    only = "lcs_rows_kernel_pipeILi13ELi4|lcs_rows_kernel_pipeILi4ELi4|lcs_long_kernelILb0"
    subprocess.check_call([sys.executable, os.path.join(CSRC, "recolor_vgprs.py"), DEV_S_FUSED, out, "--only", only, "--map", mp],
                          stdout=subprocess.DEVNULL)
    old_all, new_all, maps = open(DEV_S_FUSED).read().split("\n"), open(out).read().split("\n"), json.load(open(mp))
    assert len(maps) == 3 and RC.renaming_only(old_all, new_all) is None
    for kernel, m in maps.items():
        checked, err = RC.check_kernel(old_all, new_all, kernel, {int(k): v for k, v in m.items()})
        assert err is None and checked >= 1, (kernel, err)


def test_pass_is_a_renaming_only(recolored):
    """Outside register operands nothing changes: same lines, same opcodes, same labels; of the descriptors only
    the VGPR counts."""
    old_all, new_all, _ = recolored
    assert RC.renaming_only(old_all, new_all) is None


def test_check_catches_a_wrong_register(recolored):
    """The build step must fail on a renaming that is not one: one source register of one v_and in the bench
    kernel's loop body changed by hand."""
    old_all, new_all, maps = recolored
    k = KERNELS[0]
    bad = list(new_all)
    start = next(i for i, l in enumerate(bad) if l.startswith(k + ":"))
    seen = 0
    for i in range(start, len(bad)):
        m = re.match(r"(\s+v_and_b32_e32 v\d+, )v(\d+)(, v\d+.*)", bad[i])
        if m:
            seen += 1
            if seen == 200:
                bad[i] = m.group(1) + "v%d" % ((int(m.group(2)) + 1) % 90) + m.group(3)
                break
    assert seen == 200
    checked, err = RC.check_kernel(old_all, bad, k, {int(a): b for a, b in maps[k].items()})
    assert err is not None and "differ" in err


def _make(obj, *args):
    cmd = ["make", "-C", CSRC, f"OBJ={obj}", f"OUT={obj}/liblcsgpu.so", *args]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_build_fails_closed(tmp_path):
    """csrc/Makefile: RECOLOR=0 builds the kernels as compiled ("off"); a failing pass or a failing check falls back
    to the compiler's listing with a loud warning ("failed"); only a pass whose check succeeded is assembled ("on").
    Uses the device listing of the real build (no recompilation) in a scratch object directory."""
    if not os.path.exists(DEV_S):
        pytest.skip("famsa_amd/csrc/_obj/lcs_kernels.dev.s is made by the build (make -C famsa_amd/csrc)")
    import shutil
    obj = str(tmp_path / "obj")
    os.makedirs(obj)
    shutil.copy(DEV_S, os.path.join(obj, "lcs_kernels.dev.s"))
    final, state = os.path.join(obj, "lcs_kernels.final.s"), os.path.join(obj, "lcs_kernels.recolor.state")
    dev_text = open(DEV_S).read()

    p = _make(obj, "RECOLOR=0", final)
    assert p.returncode == 0, p.stdout
    assert open(state).read().strip() == "off" and open(final).read() == dev_text

    p = _make(obj, "RECOLOR_PASS=false", final)  # the pass itself fails (e.g. a listing it cannot parse)
    assert p.returncode == 0, p.stdout
    assert "WARNING" in p.stdout and open(state).read().strip() == "failed" and open(final).read() == dev_text

    one = "lcs_rows_kernel_pipeILi4ELi4ELi4E"  # one small kernel: the pass takes seconds
    os.utime(os.path.join(obj, "lcs_kernels.dev.s"))
    p = _make(obj, f"RECOLOR_ONLY={one}", "RECOLOR_CHECK=false", final)  # the pass runs, its check fails
    assert p.returncode == 0, p.stdout
    assert "WARNING" in p.stdout and open(state).read().strip() == "failed" and open(final).read() == dev_text

    os.utime(os.path.join(obj, "lcs_kernels.dev.s"))
    p = _make(obj, f"RECOLOR_ONLY={one}", final)  # pass + check succeed
    assert p.returncode == 0, p.stdout
    assert open(state).read().strip() == "on" and open(final).read() != dev_text
    assert "equivalent" in p.stdout

    # the plain listing assembles and links into a code object (the RECOLOR=0 product path)
    p = _make(obj, "RECOLOR=0", os.path.join(obj, "lcs_kernels.hsaco"))
    assert p.returncode == 0, p.stdout
    assert open(state).read().strip() == "off" and os.path.getsize(os.path.join(obj, "lcs_kernels.hsaco")) > 100000
