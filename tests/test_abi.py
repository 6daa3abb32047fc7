"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/lcsgpu.h declares,
reports the missing GPU loudly (no CPU fallback), and the host-only entry points work."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_bind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lcsgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lcsgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import famsa_amd
    lib = famsa_amd.load_library()
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/lcsgpu.h but not exported"


def test_library_exports_nothing_but_the_declared_symbols():
    """Built with -fvisibility=hidden and a version script: the dynamic symbol table's defined symbols are exactly the
    header's names -- no mangled internals, no kernel stubs, no libstdc++ instantiations."""
    import famsa_amd
    out = subprocess.run(["nm", "-D", "--defined-only", famsa_amd.library_path()], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) >= 3)
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))[:10]


def test_header_has_no_foreign_types():
    text = open(os.path.join(ROOT, "include", "lcsgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # declarations only
    for banned in ("torch", "hipStream_t", "std::", "#include <hip"):
        assert banned not in text


def test_version_and_device_count():
    import famsa_amd
    lib = famsa_amd.load_library()
    assert b"gfx950" in lib.lcsgpu_version()
    assert lib.lcsgpu_device_count() >= 0


def test_encode_matches_oracle_on_every_byte(oracle):
    from famsa_amd import lcsgpu
    allbytes = bytes(range(1, 256))
    assert list(lcsgpu.encode(allbytes)) == list(oracle.encode(allbytes))
    assert list(lcsgpu.encode("AR-ND")) == [0, 1, 2, 3]


def test_no_gpu_means_error_not_fallback():
    """Without a GPU the engine cannot be created; nothing computes on the CPU instead."""
    import famsa_amd
    lib = famsa_amd.load_library()
    if lib.lcsgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(famsa_amd.LcsGpuError) as e:
        famsa_amd.LcsGpu(0)
    assert "no HIP device" in str(e.value)
    # the host tool fails the same way
    from famsa_amd import hostlib as host_bind
    host = host_bind.Host()
    with pytest.raises(RuntimeError) as e2:
        host.tree_gpu(os.path.join(oracle_bind.GOLDEN, "adeno_fiber", "adeno_fiber"), "sl")
    assert "lcsgpu_create" in str(e2.value)


def test_null_and_bad_arguments_return_codes():
    import famsa_amd
    lib = famsa_amd.load_library()
    assert lib.lcsgpu_create(0, None) == -1
    assert lib.lcsgpu_destroy(None) == 0
    assert lib.lcsgpu_count(None) < 0
    n = C.c_size_t(0)
    assert lib.lcsgpu_encode(None, 3, None, C.byref(n)) == -1
    assert b"NULL" in lib.lcsgpu_last_error()


def test_bench_needs_its_gpus():
    """More ranks than GPUs without the emulation switch: an error that says so, not a hang and not a number."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and "{" not in p.stdout
