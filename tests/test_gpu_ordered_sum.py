"""famsa_amd/csrc/ordered_sum.h -- the wave that adds floats up in order, rounding like one addition after the other
(NJ's cluster sums: reference tree/NeighborJoining.cpp:44-55, 88-108) -- against a host loop, bit for bit.

The device function has no entry in the C-ABI, so the check is a small HIP program of its own
(tests/gpu_src/ordered_sum_check.hip, built by famsa_amd/csrc/Makefile into famsa_amd/_build/): 6000 vectors of
1..5000 addends of eight kinds -- plain, quantised (a tie every few additions), partly negative, like distances,
mixed magnitudes with zeros, wide exponent ranges, denormals, and rare inf / NaN / huge / -0.0f / large negative ones."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "famsa_amd", "_build", "ordered_sum_check")


@pytest.mark.gpu
def test_ordered_sums_have_the_bits_of_sequential_sums():
    assert os.path.exists(CHECK), "famsa_amd/_build/ordered_sum_check is missing: python __graft_entry__.py builds it"
    p = subprocess.run([CHECK], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    assert "OK: 0 of 6000 sums differ" in p.stdout, p.stdout
    # every path was taken: blocks of 256 at once, pieces of 64 at once, pieces added one by one
    for line in p.stdout.splitlines():
        if line.startswith("kind 3:"):
            counts = [int(w) for w in line.replace(",", " ").replace(";", " ").split() if w.isdigit()]
            assert all(c > 0 for c in counts[1:4]), line
