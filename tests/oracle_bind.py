"""ctypes bindings of the CHECKERS: oracle/liblcs_oracle.so (our C restatement) and, when it
has been built (needs /root/reference, i.e. this container), oracle/_ref/libfamsa_ref.so (the
reference's own sources).  Test infrastructure only -- nothing in famsa_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liblcs_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfamsa_ref.so")
# the reference's generators with class CLCSBP implemented over liblcsgpu.so (oracle/gpu_lcsbp.cpp): same entry points
GPUREF_SO = os.path.join(ROOT, "oracle", "_ref", "libfamsa_gpuref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_oracle():
    src = os.path.join(ROOT, "oracle", "lcs_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


class Oracle:
    def __init__(self):
        lib = C.CDLL(build_oracle())
        u32, i32, vp = C.c_uint32, C.c_int32, C.c_void_p
        lib.oracle_encode_char.restype = C.c_int
        lib.oracle_encode.restype = C.c_size_t
        lib.oracle_encode.argtypes = [C.c_char_p, C.c_size_t, vp]
        lib.oracle_lcs.restype = u32
        lib.oracle_lcs.argtypes = [vp, u32, vp, u32]
        lib.oracle_lcs_dp.restype = u32
        lib.oracle_lcs_dp.argtypes = [vp, u32, vp, u32]
        lib.oracle_lcs_rect.argtypes = [vp, vp, vp, i32, vp, i32, vp]
        lib.oracle_lcs_triangle.argtypes = [vp, vp, i32, vp]
        for name, res in [("oracle_dist_indel075_f64", C.c_double), ("oracle_dist_indel_f64", C.c_double),
                          ("oracle_dist_indel075_f32", C.c_float), ("oracle_dist_indel_f32", C.c_float),
                          ("oracle_pid_f32", C.c_float)]:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = [u32, u32, u32]
        lib.oracle_dist_triangle_f32.argtypes = [vp, vp, i32, C.c_int, vp]
        lib.oracle_format_dist.restype = C.c_int
        lib.oracle_format_dist.argtypes = [C.c_double, C.c_char_p]
        self.lib = lib

    def encode(self, residues):
        raw = residues.encode("latin-1") if isinstance(residues, str) else bytes(residues)
        out = np.empty(max(len(raw), 1), dtype=np.uint8)
        n = self.lib.oracle_encode(raw, len(raw), out.ctypes.data)
        return out[:n].copy()

    def lcs(self, ref, partner):
        ref = np.ascontiguousarray(ref, np.uint8)
        partner = np.ascontiguousarray(partner, np.uint8)
        return int(self.lib.oracle_lcs(ref.ctypes.data, len(ref), partner.ctypes.data, len(partner)))

    def lcs_dp(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return int(self.lib.oracle_lcs_dp(a.ctypes.data, len(a), b.ctypes.data, len(b)))

    def rect(self, codes, offsets, ref_ids, col_ids):
        codes = np.ascontiguousarray(codes, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        r = np.ascontiguousarray(ref_ids, np.int32)
        c = np.ascontiguousarray(col_ids, np.int32)
        out = np.empty((len(r), len(c)), np.uint32)
        self.lib.oracle_lcs_rect(codes.ctypes.data, offsets.ctypes.data, r.ctypes.data, len(r), c.ctypes.data,
                                 len(c), out.ctypes.data)
        return out

    def triangle(self, codes, offsets):
        codes = np.ascontiguousarray(codes, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        out = np.empty(n * (n - 1) // 2, np.uint32)
        self.lib.oracle_lcs_triangle(codes.ctypes.data, offsets.ctypes.data, n, out.ctypes.data)
        return out

    def dist_triangle_f32(self, lcs_triangle, lens, kind=1):
        lcs = np.ascontiguousarray(lcs_triangle, np.uint32)
        lens = np.ascontiguousarray(lens, np.uint32)
        out = np.empty(len(lcs), np.float32)
        self.lib.oracle_dist_triangle_f32(lcs.ctypes.data, lens.ctypes.data, len(lens), kind, out.ctypes.data)
        return out

    def format_dist(self, v):
        buf = C.create_string_buffer(64)
        n = self.lib.oracle_format_dist(float(v), buf)
        return buf.raw[:n].decode()


def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The reference's own code (oracle/_ref/libfamsa_ref.so); with so=GPUREF_SO the same driver and generators
    with the LCS dispatcher served by the GPU engine (GPU tests only)."""
    GT = {"sl": 0, "slink": 1, "upgma": 2, "nj": 3, "upgma_modified": 4}

    def __init__(self, so=REF_SO):
        lib = C.CDLL(so)
        vp, i = C.c_void_p, C.c_int
        lib.ref_open_fasta.restype = vp
        lib.ref_open_fasta.argtypes = [C.c_char_p]
        lib.ref_open_seqs.restype = vp
        lib.ref_open_seqs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), i]
        lib.ref_close.argtypes = [vp]
        lib.ref_count.argtypes = [vp]
        lib.ref_length.argtypes = [vp, i]
        lib.ref_codes.argtypes = [vp, i, vp]
        lib.ref_lcs_rect.argtypes = [vp, vp, i, vp, i, i, vp]
        lib.ref_tree_newick.restype = C.c_long
        lib.ref_tree_newick.argtypes = [vp, i, i, i, i, i, i, C.c_float, i, i, i, i, C.c_char_p, C.c_long]
        lib.ref_tree_newick_ex.restype = C.c_long
        lib.ref_tree_newick_ex.argtypes = [vp, i, i, i, i, i, i, C.c_float, i, i, C.c_char_p, i, i, i, C.c_char_p, C.c_long]
        lib.ref_dist_export.argtypes = [vp, i, i, i, i, i, C.c_char_p]
        lib.ref_time_triangle.restype = C.c_double
        lib.ref_time_triangle.argtypes = [vp, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
        lib.ref_clarans.argtypes = [vp, i, i, i, C.c_float, i, vp]
        self.lib = lib

    def open_fasta(self, path):
        h = self.lib.ref_open_fasta(path.encode())
        assert h, path
        return h

    def open_seqs(self, ids, residues):
        n = len(ids)
        a = (C.c_char_p * n)(*[s.encode("latin-1") for s in ids])
        b = (C.c_char_p * n)(*[s.encode("latin-1") for s in residues])
        return self.lib.ref_open_seqs(a, b, n)

    def close(self, h):
        self.lib.ref_close(h)

    def codes(self, h):
        out = []
        for k in range(self.lib.ref_count(h)):
            a = np.empty(max(self.lib.ref_length(h, k), 1), np.uint8)
            self.lib.ref_codes(h, k, a.ctypes.data)
            out.append(a[: self.lib.ref_length(h, k)].copy())
        return out

    def lcs_rect(self, h, ref_ids, col_ids, isa=2):
        r = np.ascontiguousarray(ref_ids, np.int32)
        c = np.ascontiguousarray(col_ids, np.int32)
        out = np.empty((len(r), len(c)), np.uint32)
        self.lib.ref_lcs_rect(h, r.ctypes.data, len(r), c.ctypes.data, len(c), isa, out.ctypes.data)
        return out

    def tree(self, h, gt, distance=1, heuristic=0, subtree=0, sample=0, threshold=0, cluster_fraction=0.0,
             cluster_iters=0, keep_dups=0, threads=4, isa=2, cap=1 << 25, num_evals=0, dump_seeds=None):
        buf = C.create_string_buffer(cap)
        if num_evals or dump_seeds:
            n = self.lib.ref_tree_newick_ex(h, self.GT[gt], distance, heuristic, subtree, sample, threshold, cluster_fraction,
                                            cluster_iters, num_evals, dump_seeds.encode() if dump_seeds else None,
                                            keep_dups, threads, isa, buf, len(buf))
        else:
            n = self.lib.ref_tree_newick(h, self.GT[gt], distance, heuristic, subtree, sample, threshold,
                                         cluster_fraction, cluster_iters, keep_dups, threads, isa, buf, len(buf))
        assert n >= 0, n
        return buf.raw[:n]

    def dist_export(self, h, path, distance=1, square=False, pid=False, threads=4, isa=2):
        rc = self.lib.ref_dist_export(h, distance, int(square), int(pid), threads, isa, path.encode())
        assert rc == 0

    def clarans(self, triangle, n_elems, n_medoids, n_fixed=1, explore_fraction=0.1, num_local=2):
        tri = np.ascontiguousarray(triangle, np.float32)
        out = np.zeros(n_medoids, np.int32)
        self.lib.ref_clarans(tri.ctypes.data, n_elems, n_medoids, n_fixed, explore_fraction, num_local, out.ctypes.data)
        return out

    def time_triangle(self, h, n_use, threads, isa=2, want_matrix=False):
        pairs, cells = C.c_double(0), C.c_double(0)
        mat = None
        if want_matrix:
            mat = np.empty(n_use * (n_use - 1) // 2, np.float32)
        sec = self.lib.ref_time_triangle(h, n_use, threads, isa, C.byref(pairs), C.byref(cells),
                                         mat.ctypes.data if mat is not None else None)
        return sec, pairs.value, cells.value, mat
