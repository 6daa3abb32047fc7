"""ctypes binding of libfamsa_host.so -- the C++ host layer above the C-ABI (working order, guide-tree
builders over GPU LCS, Newick / CSV writers) -- plus two convenience functions for Python callers."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(_HERE, "libfamsa_host.so")
CLI = os.path.join(_HERE, "famsa-gpu")
DIST = {"indel_div_lcs": 0, "indel075_div_lcs": 1}


class Host:
    def __init__(self):
        from .lcsgpu import load_library
        load_library()  # resolves liblcsgpu.so (and lets torch's HIP runtime load first)
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} not found: build it with `make -C famsa_amd/host`")
        lib = C.CDLL(HOST_SO)
        lib.famsa_host_last_error.restype = C.c_char_p
        lib.famsa_host_tree_from_matrix.restype = C.c_long
        heur = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        lib.famsa_host_tree_from_matrix.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, *heur,
                                                    C.c_char_p, C.c_long]
        lib.famsa_host_tree_from_matrix_ex.restype = C.c_long
        lib.famsa_host_tree_from_matrix_ex.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, *heur, C.c_int,
                                                       C.c_uint32, C.c_char_p, C.c_char_p, C.c_long]
        lib.famsa_host_dist_export_from_matrix.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                           C.c_char_p]
        lib.famsa_host_tree_gpu.restype = C.c_long
        lib.famsa_host_tree_gpu.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, *heur, C.c_char_p,
                                            C.c_long]
        lib.famsa_host_dist_export_gpu.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
        lib.famsa_host_workset.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.famsa_host_format_distance.argtypes = [C.c_double, C.c_char_p]
        lib.famsa_host_records.restype = C.c_long
        lib.famsa_host_records.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                           C.c_long]
        lib.famsa_host_clarans.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        lib.famsa_host_newick.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p,
                                          C.c_long]
        lib.famsa_host_newick.restype = C.c_long
        self.lib = lib

    def _err(self):
        return RuntimeError(self.lib.famsa_host_last_error().decode())

    HEUR = {None: 0, "parttree": 1, "medoidtree": 2}

    def _text(self, call, cap=1 << 25):
        """Run a text-returning entry point; -1 = error, <= -2 = the buffer must hold -(n + 1) bytes: retry."""
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            n = call(buf)
            if n >= 0:
                return buf.raw[:n]
            if n == -1:
                raise self._err()
            cap = -(n + 1)
        raise self._err()

    def tree_from_matrix(self, fasta, square, method, distance="indel075_div_lcs", keep_duplicates=False,
                         heuristic=None, subtree_size=0, sample_size=0, threshold=0, cluster_fraction=0.0,
                         cluster_iters=0, num_evals=0, chained_seed=0, dump_seeds=None):
        """`-gt <method> -gt_export` over a caller-supplied oriented LCS matrix; num_evals / chained_seed / dump_seeds are the
        CLI's -num_evals, the seed of -gt chained, and -dump_seeds <file>."""
        sq = np.ascontiguousarray(square, np.uint32)
        return self._text(lambda buf: self.lib.famsa_host_tree_from_matrix_ex(
            fasta.encode(), sq.ctypes.data, method.encode(), DIST[distance], int(keep_duplicates), self.HEUR[heuristic],
            subtree_size, sample_size, threshold, cluster_fraction, cluster_iters, num_evals, chained_seed,
            dump_seeds.encode() if dump_seeds else None, buf, len(buf)))

    def dist_export_from_matrix(self, fasta, square, path, distance="indel075_div_lcs", square_matrix=False,
                                pid=False):
        sq = np.ascontiguousarray(square, np.uint32)
        if self.lib.famsa_host_dist_export_from_matrix(fasta.encode(), sq.ctypes.data, DIST[distance],
                                                       int(square_matrix), int(pid), path.encode()) != 0:
            raise self._err()

    def tree_gpu(self, fasta, method, distance="indel075_div_lcs", keep_duplicates=False, device=0, heuristic=None,
                 subtree_size=0, sample_size=0, threshold=0, cluster_fraction=0.0, cluster_iters=0):
        return self._text(lambda buf: self.lib.famsa_host_tree_gpu(
            fasta.encode(), device, method.encode(), DIST[distance], int(keep_duplicates), self.HEUR[heuristic],
            subtree_size, sample_size, threshold, cluster_fraction, cluster_iters, buf, len(buf)))

    def dist_export_gpu(self, fasta, path, distance="indel075_div_lcs", square_matrix=False, pid=False, device=0):
        if self.lib.famsa_host_dist_export_gpu(fasta.encode(), device, DIST[distance], int(square_matrix), int(pid),
                                               path.encode()) != 0:
            raise self._err()

    def workset(self, fasta, n, keep_duplicates=False):
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        u = self.lib.famsa_host_workset(fasta.encode(), int(keep_duplicates), a.ctypes.data, b.ctypes.data, n)
        if u < 0:
            raise self._err()
        return u, a, b

    def newick(self, left, right, names, sorted2unique=None):
        """Newick text of the tree whose internal nodes (children left[i], right[i], leaves first) follow the leaves; with
        sorted2unique the duplicates are re-attached first (GuideTree::fromUnique) and names has one entry per record."""
        left = np.ascontiguousarray(left, np.int32)
        right = np.ascontiguousarray(right, np.int32)
        n_leaves = len(left) + 1
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        s2u = None if sorted2unique is None else np.ascontiguousarray(sorted2unique, np.int32)
        cap = 64 + sum(len(n) + 12 for n in names)
        return self._text(lambda buf: self.lib.famsa_host_newick(
            left.ctypes.data, right.ctypes.data, n_leaves, len(left), C.cast(arr, C.c_void_p), len(names),
            None if s2u is None else s2u.ctypes.data, buf, len(buf)), cap=cap)

    def format_distance(self, v):
        buf = C.create_string_buffer(64)
        n = self.lib.famsa_host_format_distance(float(v), buf)
        return buf.raw[:n].decode()

    def clarans(self, triangle, n_elems, n_medoids, n_fixed=1, explore_fraction=0.1, num_local=2):
        tri = np.ascontiguousarray(triangle, np.float32)
        out = np.zeros(n_medoids, np.int32)
        if self.lib.famsa_host_clarans(tri.ctypes.data, n_elems, n_medoids, n_fixed, explore_fraction, num_local,
                                       out.ctypes.data) != 0:
            raise self._err()
        return out

    def records(self, fasta, n_threads=0):
        """(ids, [code arrays]) as the FASTA reader delivers them."""
        size = os.path.getsize(fasta) + 16
        for attempt in range(4):  # a compressed file needs more room than its size on disk
            ids = C.create_string_buffer(size)
            codes = np.zeros(size, np.uint8)
            offs = np.zeros(size // 2 + 2, np.uint64)
            n = self.lib.famsa_host_records(fasta.encode(), n_threads, ids, size, codes.ctypes.data, size,
                                            offs.ctypes.data, len(offs))
            if n >= 0 or b"buffer too small" not in self.lib.famsa_host_last_error():
                break
            size *= 8
        if n < 0:
            raise self._err()
        names = ids.value.decode("latin-1").split("\n")[:-1] if n else []
        return names, [codes[int(offs[i]):int(offs[i + 1])].copy() for i in range(n)]


_HOST = None


def _host():
    global _HOST
    if _HOST is None:
        _HOST = Host()
    return _HOST


def guide_tree(fasta, gt="sl", distance="indel075_div_lcs", keep_duplicates=False, heuristic=None, device=0, **medoid):
    """Newick text (bytes) of `famsa -gt <gt> -gt_export <fasta>`, the LCS and the tree reducers on GPU `device`.
    heuristic: None, "medoidtree" or "parttree" (keyword arguments subtree_size, sample_size, threshold,
    cluster_fraction, cluster_iters as in FAMSA)."""
    return _host().tree_gpu(fasta, gt, distance, keep_duplicates, device, heuristic, **medoid)


def dist_export(fasta, csv_path, distance="indel075_div_lcs", square_matrix=False, pid=False, device=0):
    """`famsa -dist_export [-square_matrix] [-pid] <fasta> <csv_path>` with the LCS on GPU `device`."""
    _host().dist_export_gpu(fasta, csv_path, distance, square_matrix, pid, device)
