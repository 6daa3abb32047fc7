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
GT = {"sl": 0, "upgma": 2, "nj": 3, "upgma_modified": 4}


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
    ("adeno_fiber/adeno_fiber", "sl", "adeno_fiber/sl.dnd"),  # the default method: lcsgpu_mst_prim -> the reference's mst_to_dendogram
    ("adeno_fiber_duplicates/adeno_fiber_duplicates", "sl", "adeno_fiber_duplicates/sl.dnd"),
    ("hemopexin/hemopexin", "sl", "hemopexin/sl.dnd"),
    ("adversarial_tree.fasta", "sl", "adversarial_tree_sl.dnd"),  # orientation-sensitive refs: the step-by-step Prim kernel
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


# ---- the dispatcher seam: class CLCSBP implemented over liblcsgpu.so (oracle/gpu_lcsbp.cpp) ------------------------
# oracle/_ref/libfamsa_gpuref.so = the reference's six generators (object code from /root/reference) + ref_harness.cpp
# + gpu_lcsbp.cpp INSTEAD of the reference's lcs/lcsbp.cpp, lcsbp_classic.cpp and simd/*: it holds no CPU LCS code, so
# whatever it returns was computed from GPU LCS lengths by the reference's own batch templates, Transform functors,
# tree algorithms, CLARANS and writers.

@pytest.fixture(scope="module")
def gpuref():
    famsa_amd.load_library()
    assert os.path.exists(oracle_bind.GPUREF_SO), "build it: make -C oracle ref (needs /root/reference; the GPU box uses the prebuilt one)"
    ref = oracle_bind.Ref(oracle_bind.GPUREF_SO)
    ref.lib.gpu_lcsbp_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

    def stats():
        rows, values = C.c_uint64(0), C.c_uint64(0)
        ref.lib.gpu_lcsbp_stats(C.byref(rows), C.byref(values))
        return rows.value, values.value
    ref.stats = stats
    return ref


def test_gpuref_has_no_cpu_lcs_code():
    import subprocess
    syms = subprocess.run(["nm", "-DC", oracle_bind.GPUREF_SO], stdout=subprocess.PIPE, text=True).stdout
    assert "CLCSBP::GetLCSBP" in syms and "lcsgpu_lcs_rect" in syms
    assert "CLCSBP_Classic" not in syms and "CLCSBP_AVX" not in syms


@pytest.mark.parametrize("case,gt,gold", [
    ("adeno_fiber/adeno_fiber", "sl", "adeno_fiber/sl.dnd"),
    ("adeno_fiber/adeno_fiber", "slink", "adeno_fiber/slink.dnd"),
    ("adeno_fiber/adeno_fiber", "upgma", "adeno_fiber/upgma.dnd"),
    ("adeno_fiber/adeno_fiber", "nj", "adeno_fiber/nj.dnd"),
    ("adeno_fiber_duplicates/adeno_fiber_duplicates", "sl", "adeno_fiber_duplicates/sl.dnd"),
    ("hemopexin/hemopexin", "sl", "hemopexin/sl.dnd"),
    ("hemopexin/hemopexin", "slink", "hemopexin/slink.dnd"),
    ("hemopexin/hemopexin", "upgma", "hemopexin/upgma.dnd"),
    ("adversarial_tree.fasta", "sl", "adversarial_tree_sl.dnd"),      # carry-quirk refs: MSTPrim's own orientation
    ("adversarial_tree.fasta", "slink", "adversarial_tree_slink.dnd"),  # ... and the row orientation
    ("adversarial_tree.fasta", "upgma", "adversarial_tree_upgma.dnd"),
    ("adversarial_tree.fasta", "nj", "adversarial_tree_nj.dnd"),
])
def test_reference_generators_at_the_dispatcher_seam(gpuref, case, gt, gold):
    """MSTPrim::run_view, SingleLinkage::run, UPGMA::run, NeighborJoining::run -- the reference's object code --
    with every GetLCSBP group answered by the GPU engine."""
    fasta = os.path.join(G, case)
    rows0, values0 = gpuref.stats()
    h = gpuref.open_fasta(fasta)
    got = gpuref.tree(h, gt, threads=4)
    gpuref.close(h)
    rows1, values1 = gpuref.stats()
    assert rows1 > rows0 and values1 > values0  # the generator really was served from the engine
    assert got == open(os.path.join(G, gold), "rb").read()
    assert got == famsa_amd.guide_tree(fasta, gt)


@pytest.mark.parametrize("gt,params,gold", [
    ("sl", {}, "medoid-sl.dnd"),
    ("upgma", {}, "medoid-upgma.dnd"),
    ("nj", {}, "medoid-nj.dnd"),
    ("slink", dict(subtree=10, sample=100, threshold=100, cluster_fraction=0.2, cluster_iters=1), "medoid-slink-params.dnd"),
])
def test_reference_medoid_tree_at_the_dispatcher_seam(gpuref, gt, params, gold):
    """FastTree::doStep / makeEvaluation / clusterSeeds + CLARANS + the partial generators (reference object code):
    seeds x all rectangles, sample matrices and leaf matrices all from the GPU; the upstream medoid goldens."""
    h = gpuref.open_fasta(os.path.join(G, "hemopexin", "hemopexin"))
    got = gpuref.tree(h, gt, heuristic=2, threads=4, **params)
    gpuref.close(h)
    assert got == open(os.path.join(G, "hemopexin", gold), "rb").read()


@pytest.mark.parametrize("square,pid,gold", [(False, False, "dist.csv"), (True, False, "dist_sq.csv"),
                                             (False, True, "pid.csv"), (True, True, "pid_sq.csv")])
def test_reference_dist_export_at_the_dispatcher_seam(gpuref, tmp_path, square, pid, gold):
    """DistanceCalculator::run (reference object code: row workers + its CSV writer) over GPU LCS lengths:
    byte-identical to the upstream CSV goldens and to famsa-gpu's own writer."""
    fasta = os.path.join(G, "adeno_fiber", "adeno_fiber")
    h = gpuref.open_fasta(fasta)
    out = str(tmp_path / "d.csv")
    gpuref.dist_export(h, out, square=square, pid=pid, threads=4)
    gpuref.close(h)
    got = open(out, "rb").read()
    assert got == open(os.path.join(G, "adeno_fiber", gold), "rb").read()
    out2 = str(tmp_path / "e.csv")
    famsa_amd.dist_export(fasta, out2, square_matrix=square, pid=pid)
    assert got == open(out2, "rb").read()


def test_reference_dist_export_hemopexin_at_the_dispatcher_seam(gpuref, tmp_path):
    import hashlib
    import json
    meta = json.load(open(os.path.join(G, "meta.json")))["hemopexin"]
    h = gpuref.open_fasta(os.path.join(G, "hemopexin", "hemopexin"))
    out = str(tmp_path / "d.csv")
    gpuref.dist_export(h, out, threads=8)
    gpuref.close(h)
    assert os.path.getsize(out) == meta["dist_csv_bytes"]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == meta["dist_csv_sha256"]


def test_raw_lcs_through_getlcsbp(gpuref):
    """CLCSBP::GetLCSBP itself, groups of 8 with a null-padded tail as calculateDistanceVector issues them
    (ref_harness.cpp: ref_lcs_rect), on the adversarial set (carry quirks, word-boundary lengths, non-matching
    symbols, all-X, duplicates): the matrix the reference's own CLCSBP produced (tests/golden/adversarial_lcs.npz)."""
    import numpy as np
    want = np.load(os.path.join(G, "adversarial_lcs.npz"))["classic"]  # == "avx2" (meta.json: classic_eq_avx2)
    h = gpuref.open_fasta(os.path.join(G, "adversarial.fasta"))
    n = gpuref.lib.ref_count(h)
    got = gpuref.lcs_rect(h, np.arange(n), np.arange(n))
    gpuref.close(h)
    assert got.shape == want.shape and (got == want).all()
