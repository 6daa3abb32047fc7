"""lcsgpu_nj's two ways to run the merges of NeighborJoining::computeTree (reference tree/NeighborJoining.cpp:33-118) --
inside ONE resident launch (nj_loop_kernels.hip: owned blocks of the triangle, tagged words between the workgroups, the
sum of the merged cluster added up beside the scan, the triangle squeezed from time to time) and four launches per merge
(tree_kernels.hip) -- must give the same tree on every shape: the launches are the form the goldens were pinned with in
rounds 1-5, and the reference library checks both on the sets it can do in seconds (tests/test_gpu_endtoend.py)."""
import os

import numpy as np
import pytest

from famsa_amd import seqio
from famsa_amd.lcsgpu import LcsGpuError

pytestmark = pytest.mark.gpu


def _family(n, length, seed):
    return seqio.synth_family(n, length, seed=seed)


def _short(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(0, 4, size=int(rng.integers(5, 30))).astype(np.uint8) for _ in range(n)]  # many equal distances


def _nj(engine, kind, tune):
    """(left, right), or the text of the error (a pair with LCS 0 makes the reference's result degenerate: both ways say so)"""
    old = os.environ.get("LCSGPU_TUNE")
    os.environ["LCSGPU_TUNE"] = tune  # (read by the library at every call)
    try:
        return engine.nj(kind)
    except LcsGpuError as e:
        assert "degenerate" in str(e), e
        return "degenerate"
    finally:
        if old is None:
            del os.environ["LCSGPU_TUNE"]
        else:
            os.environ["LCSGPU_TUNE"] = old


@pytest.mark.parametrize("n", [3, 4, 5, 17, 64, 65, 257, 300, 700, 1500, 2600])
def test_resident_launch_equals_launches_per_merge(engine, n):
    for seqs, kind in ((_family(n, 120, seed=n), 1), (_short(n, seed=n + 1), 1), (_family(n, 90, seed=3 * n), 0)):
        engine.upload_seqs(seqs)
        want = _nj(engine, kind, "nj_loop=0")
        got = _nj(engine, kind, "nj_loop=1")
        if isinstance(want, str) or isinstance(got, str):
            assert isinstance(want, str) and isinstance(got, str)
            continue
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()


@pytest.mark.parametrize("tune", ["nj_loop=1,nj_squeeze_min=0", "nj_loop=1,nj_groups=7", "nj_loop=1,nj_groups=1",
                                  "nj_loop=1,nj_groups=64,nj_squeeze_min=100000"])
def test_shapes_of_the_resident_launch(engine, tune):
    """The triangle squeezed as early as the rule allows, never, and odd numbers of workgroups (the ownership of the
    triangle's blocks, the exchange of candidates and the spread of the updates all follow the count)."""
    for n in (40, 333, 1200):
        seqs = _family(n, 100, seed=7 * n)
        engine.upload_seqs(seqs)
        want = _nj(engine, 1, "nj_loop=0")
        got = _nj(engine, 1, tune)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()


def test_a_resident_launch_that_cannot_meet_gives_up_and_the_merges_run_as_launches(engine):
    """Twice as many workgroups as CUs: the resident half waits for words that cannot come, gives up after ~1 s (nothing hangs),
    and lcsgpu_nj runs the merges as launches from the same LCS values -- what happens when another process holds CUs."""
    seqs = _family(500, 100, seed=11)
    engine.upload_seqs(seqs)
    want = _nj(engine, 1, "nj_loop=0")
    got = _nj(engine, 1, "nj_loop=1,nj_oversubscribe=1")
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
