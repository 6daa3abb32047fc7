"""famsa_amd -- MI355X (gfx950) engine for FAMSA's all-pairs bit-parallel LCS hot path.

The product is the C-ABI shared library ``liblcsgpu.so`` (``include/lcsgpu.h``; sources in
``famsa_amd/csrc``) and the C++ host tools in ``famsa_amd/host``.  This Python package is a
thin ctypes binding used by the tests and ``bench.py`` (``LcsGpu`` = the C-ABI, ``guide_tree`` /
``dist_export`` = the host layer); it never computes an LCS itself and raises if the HIP library is missing.
"""
from .lcsgpu import LcsGpu, LcsGpuGroup, LcsGpuError, load_library, library_path  # noqa: F401
from . import seqio  # noqa: F401
from .hostlib import guide_tree, dist_export  # noqa: F401,E402
