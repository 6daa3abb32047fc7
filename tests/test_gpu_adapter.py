"""The drop-in claim, compiled (oracle/adapter_check.cpp): the REFERENCE's own UPGMA<D>::computeTree and
NeighborJoining<D>::computeTree -- object code built from /root/reference by oracle/Makefile -- fed by a
GpuDistanceProvider that sits where AbstractTreeGenerator::calculateDistanceMatrix sits
(reference tree/AbstractTreeGenerator.hpp:378-398) and takes its LCS lengths from liblcsgpu.so.  The Newick it
yields must equal the reference's goldens and what famsa-gpu's own host layer yields."""
import ctypes as C
import os

import pytest

import famsa_amd
import oracle_bind

pytestmark = pytest.mark.gpu
G = oracle_bind.GOLDEN
ADAPTER_SO = os.path.join(os.path.dirname(oracle_bind.REF_SO), "libfamsa_adapter.so")
GT = {"upgma": 2, "nj": 3, "upgma_modified": 4}


@pytest.fixture(scope="module")
def adapter():
    famsa_amd.load_library()
    assert os.path.exists(ADAPTER_SO), "build it: make -C oracle ref (needs /root/reference; the GPU box uses the prebuilt one)"
    ref = oracle_bind.Ref()
    lib = C.CDLL(ADAPTER_SO)
    lib.adapter_tree_newick.restype = C.c_long
    lib.adapter_tree_newick.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_long]

    def run(fasta, gt, distance=1, keep_dups=0):
        h = ref.open_fasta(fasta)
        buf = C.create_string_buffer(1 << 24)
        n = lib.adapter_tree_newick(h, GT[gt], distance, keep_dups, 0, buf, len(buf))
        ref.close(h)
        assert n >= 0, n
        return buf.raw[:n]
    return run


@pytest.mark.parametrize("case,gt,gold", [
    ("adeno_fiber/adeno_fiber", "upgma", "adeno_fiber/upgma.dnd"),
    ("adeno_fiber/adeno_fiber", "nj", "adeno_fiber/nj.dnd"),
    ("adeno_fiber_duplicates/adeno_fiber_duplicates", "upgma", None),
    ("hemopexin/hemopexin", "upgma", "hemopexin/upgma.dnd"),
    ("adversarial_tree.fasta", "upgma", "adversarial_tree_upgma.dnd"),
    ("adversarial_tree.fasta", "nj", "adversarial_tree_nj.dnd"),
    ("adversarial_tree.fasta", "upgma_modified", None),
])
def test_reference_generators_over_gpu_distances(adapter, case, gt, gold):
    fasta = os.path.join(G, case)
    got = adapter(fasta, gt)
    if gold:
        assert got == open(os.path.join(G, gold), "rb").read()
    assert got == famsa_amd.guide_tree(fasta, gt)  # the re-implemented host layer (device reducers) agrees


def test_other_distance_and_kept_duplicates(adapter):
    fasta = os.path.join(G, "adeno_fiber_duplicates", "adeno_fiber_duplicates")
    assert adapter(fasta, "upgma", distance=0, keep_dups=1) == \
        famsa_amd.guide_tree(fasta, "upgma", distance="indel_div_lcs", keep_duplicates=True)
