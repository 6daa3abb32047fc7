"""ctypes binding of include/lcsgpu.h (one-to-one; no compute here)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LcsGpuError(RuntimeError):
    pass


def library_path():
    return os.environ.get("LCSGPU_LIB") or os.path.join(_HERE, "liblcsgpu.so")


def load_library():
    """Load liblcsgpu.so from the package directory.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LcsGpuError(
            f"{path} not found: build it with `make -C famsa_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback")
    # torch wheels carry their own libamdhip64; two HIP runtimes in one process do not coexist, so
    # when torch is installed let it load its copy first and liblcsgpu.so bind to that one.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    pi32 = C.POINTER(C.c_int32)
    sig = {
        "lcsgpu_version": (C.c_char_p, []),
        "lcsgpu_last_error": (C.c_char_p, []),
        "lcsgpu_device_count": (C.c_int, []),
        "lcsgpu_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "lcsgpu_destroy": (C.c_int, [vp]),
        "lcsgpu_reserve_lanes": (C.c_int, [vp, i32]),
        "lcsgpu_encode": (C.c_int, [C.c_char_p, sz, vp, C.POINTER(sz)]),
        "lcsgpu_upload": (C.c_int, [vp, vp, vp, i32]),
        "lcsgpu_upload_ordered": (C.c_int, [vp, vp, vp, i32, vp, i32]),
        "lcsgpu_count": (i32, [vp]),
        "lcsgpu_length": (i32, [vp, i32]),
        "lcsgpu_orientation_flags": (i32, [vp, vp]),
        "lcsgpu_lcs_rect": (C.c_int, [vp, pi32, i32, i32, pi32, i32, i32, vp, i64, C.c_int]),
        "lcsgpu_lcs_rect_dev": (C.c_int, [vp, pi32, i32, i32, pi32, i32, i32, vp, i64, C.c_int, C.c_int]),
        "lcsgpu_lcs_triangle": (C.c_int, [vp, i32, i32, vp, C.c_int]),
        "lcsgpu_lcs_triangle_dev": (C.c_int, [vp, i32, i32, vp, C.c_int, C.c_int]),
        "lcsgpu_lcs_triangle_ids": (C.c_int, [vp, pi32, i32, vp, C.c_int]),
        "lcsgpu_row_minima_dev": (C.c_int, [vp, vp, C.c_int, i32, i32, C.c_int, vp, C.c_int]),
        "lcsgpu_mst_prim": (C.c_int, [vp, C.c_int, vp]),
        "lcsgpu_mst_shard_begin": (C.c_int, [vp, vp, C.c_int, i32, i32, C.c_int]),
        "lcsgpu_mst_shard_best": (C.c_int, [vp, vp, vp]),
        "lcsgpu_mst_shard_merge": (C.c_int, [vp, vp, i32, pi32]),
        "lcsgpu_mst_shard_finish": (C.c_int, [vp, vp]),
        "lcsgpu_mst_shard_edges": (C.c_int, [vp, vp]),
        "lcsgpu_mst_merge_host": (C.c_int, [vp, i32, i32, vp, vp, pi32]),
        "lcsgpu_mst_shard_set_components": (C.c_int, [vp, vp]),
        "lcsgpu_mst_order_edges": (C.c_int, [vp, i32]),
        "lcsgpu_multi_lcs_triangle": (C.c_int, [vp, i32, i32, i32, vp, C.c_int]),
        "lcsgpu_multi_upgma": (C.c_int, [vp, i32, C.c_int, C.c_int, vp, vp]),
        "lcsgpu_multi_nj": (C.c_int, [vp, i32, C.c_int, vp, vp]),
        "lcsgpu_multi_mst_prim": (C.c_int, [vp, i32, C.c_int, vp]),
        "lcsgpu_multi_transport": (C.c_int, [vp, i32, C.c_char_p, sz]),
        "lcsgpu_upgma": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
        "lcsgpu_nj": (C.c_int, [vp, C.c_int, vp, vp]),
        "lcsgpu_lcs_triangles_batch": (C.c_int, [vp, pi32, C.POINTER(C.c_int64), i32, vp, C.c_int]),
        "lcsgpu_assign_seeds": (C.c_int, [vp, pi32, i32, pi32, i32, C.c_int, i32, vp, vp]),
        "lcsgpu_clarans": (C.c_int, [vp, pi32, i32, C.c_int, i32, i32, C.c_float, i32, pi32]),
        "lcsgpu_clarans_batch": (C.c_int, [vp, pi32, C.POINTER(C.c_int64), i32, C.c_int, pi32, i32, C.c_float, i32, pi32]),
        "lcsgpu_assign_seeds_batch": (C.c_int, [vp, pi32, C.POINTER(C.c_int64), pi32, C.POINTER(C.c_int64), i32, C.c_int, vp, vp]),
        "lcsgpu_dist_text_begin": (C.c_int, [vp, C.c_char_p, vp, C.c_int, C.c_int, i32]),
        "lcsgpu_dist_text_submit": (C.c_int, [vp, i32, i32, i32]),
        "lcsgpu_dist_text_wait": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "lcsgpu_dist_text_end": (C.c_int, [vp]),
        "lcsgpu_sync": (C.c_int, [vp]),
        "lcsgpu_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i32)]),
        "lcsgpu_total_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "lcsgpu_stream": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


MST_EDGE = np.dtype([("from", np.int32), ("to", np.int32), ("dist", np.float64)])  # lcsgpu_mst_edge
MST_KEY = np.dtype([("dist_bits", np.uint64), ("id", np.uint64)])                  # lcsgpu_mst_key
MST_TRIANGLE_ORIENTATION = 0x100
MST_COMPUTE = 0x200  # lcsgpu_mst_shard_begin: the LCS launch itself does the local half (see include/lcsgpu.h)


def mst_merge_host(keys, comp, edges, n_edges):
    """lcsgpu_mst_merge_host: keys [n_parts, n] MST_KEY (host), comp int32[n] and edges MST_EDGE[n-1]
    updated in place; returns the new edge count.  Pure host code -- works without a GPU."""
    lib = load_library()
    keys = np.ascontiguousarray(keys, dtype=MST_KEY)
    n_parts, n = keys.shape
    assert comp.dtype == np.int32 and comp.flags.c_contiguous and len(comp) == n
    assert edges.dtype == MST_EDGE and edges.flags.c_contiguous and len(edges) >= max(n - 1, 0)
    cnt = C.c_int32(int(n_edges))
    rc = lib.lcsgpu_mst_merge_host(keys.ctypes.data, n_parts, n, comp.ctypes.data, edges.ctypes.data, C.byref(cnt))
    if rc:
        raise LcsGpuError(f"lcsgpu error {rc}: {lib.lcsgpu_last_error().decode()}")
    return cnt.value


def mst_order_edges(edges, n):
    """lcsgpu_mst_order_edges: in place, into Prim's insertion order from vertex 0.  Pure host code."""
    lib = load_library()
    assert edges.dtype == MST_EDGE and edges.flags.c_contiguous and len(edges) == max(n - 1, 0)
    rc = lib.lcsgpu_mst_order_edges(edges.ctypes.data if len(edges) else None, n)
    if rc:
        raise LcsGpuError(f"lcsgpu error {rc}: {lib.lcsgpu_last_error().decode()}")
    return edges


def _ids(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=np.int32)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_int32))


def encode(residues):
    """Residue string -> uint8 symbol codes through the library's encoder (lcsgpu_encode)."""
    lib = load_library()
    raw = residues.encode("latin-1") if isinstance(residues, str) else bytes(residues)
    out = np.empty(max(len(raw), 1), dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib.lcsgpu_encode(raw, len(raw), out.ctypes.data, C.byref(n))
    if rc:
        raise LcsGpuError(lib.lcsgpu_last_error().decode())
    return out[: n.value].copy()


class LcsGpu:
    """One engine context on one GPU (mirrors `CLCSBP` + the batch-distance templates)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        self._check(self._lib.lcsgpu_create(int(device), C.byref(self._ctx)))
        self.n = 0
        self.lengths = np.zeros(0, dtype=np.uint32)

    def _check(self, rc):
        if rc != 0:
            raise LcsGpuError(f"lcsgpu error {rc}: {self._lib.lcsgpu_last_error().decode()}")

    def close(self):
        if self._ctx:
            self._lib.lcsgpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        self._check(self._lib.lcsgpu_upload(self._ctx, codes.ctypes.data if codes.size else None,
                                            offsets.ctypes.data, n))
        self.n = n
        self.lengths = np.diff(offsets.astype(np.int64)).astype(np.uint32)

    def upload_ordered(self, codes, offsets, order):
        """Sequence k of the set = record order[k] of (codes, offsets); records not named are left out."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._check(self._lib.lcsgpu_upload_ordered(self._ctx, codes.ctypes.data if codes.size else None, offsets.ctypes.data,
                                                    len(offsets) - 1, order.ctypes.data, len(order)))
        self.n = len(order)
        self.lengths = np.diff(offsets.astype(np.int64)).astype(np.uint32)[order] if len(order) else np.zeros(0, np.uint32)

    def upload_seqs(self, seqs):
        """seqs: list of uint8 code arrays."""
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        codes = np.concatenate(seqs) if len(seqs) and offsets[-1] > 0 else np.zeros(0, dtype=np.uint8)
        self.upload(codes, offsets)

    def orientation_flags(self):
        flags = np.zeros(max(self.n, 1), dtype=np.uint8)
        rc = self._lib.lcsgpu_orientation_flags(self._ctx, flags.ctypes.data)
        if rc < 0:
            self._check(rc)
        return flags[: self.n]

    def lcs_rect(self, refs, cols, dtype=np.uint16):
        """refs / cols: (begin, count) tuple or id array.  Returns [n_refs, n_cols] host array."""
        r_arr, r_ptr, r_begin, n_refs = self._spec(refs)
        c_arr, c_ptr, c_begin, n_cols = self._spec(cols)
        out = np.empty((n_refs, n_cols), dtype=dtype)
        self._check(self._lib.lcsgpu_lcs_rect(self._ctx, r_ptr, r_begin, n_refs, c_ptr, c_begin, n_cols,
                                              out.ctypes.data if out.size else None, n_cols, out.itemsize))
        return out

    def lcs_rect_dev(self, refs, cols, d_out_ptr, ld, elem_size, sync=False):
        r_arr, r_ptr, r_begin, n_refs = self._spec(refs)
        c_arr, c_ptr, c_begin, n_cols = self._spec(cols)
        self._check(self._lib.lcsgpu_lcs_rect_dev(self._ctx, r_ptr, r_begin, n_refs, c_ptr, c_begin, n_cols,
                                                  C.c_void_p(d_out_ptr), ld, elem_size, 1 if sync else 0))

    def lcs_triangle(self, row_begin=0, row_end=None, dtype=np.uint16):
        row_end = self.n if row_end is None else row_end
        count = row_end * (row_end - 1) // 2 - row_begin * (row_begin - 1) // 2
        out = np.empty(max(count, 0), dtype=dtype)
        self._check(self._lib.lcsgpu_lcs_triangle(self._ctx, row_begin, row_end,
                                                  out.ctypes.data if out.size else None, out.itemsize))
        return out

    def lcs_triangle_ids(self, ids, dtype=np.uint16):
        arr, ptr = _ids(ids)
        n = len(arr)
        out = np.empty(max(n * (n - 1) // 2, 0), dtype=dtype)
        self._check(self._lib.lcsgpu_lcs_triangle_ids(self._ctx, ptr, n, out.ctypes.data if out.size else None,
                                                      out.itemsize))
        return out

    def lcs_triangles_batch(self, groups, dtype=np.uint16):
        """Packed lower triangles of several id lists, one call; returns a list of 1-D arrays."""
        sizes = [len(g) for g in groups]
        offs = np.zeros(len(groups) + 1, dtype=np.int64)
        np.cumsum(sizes, out=offs[1:])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(g, np.int32) for g in groups])
                                   if offs[-1] else np.zeros(0, np.int32), dtype=np.int32)
        tri = [m * (m - 1) // 2 for m in sizes]
        base = np.zeros(len(groups) + 1, dtype=np.int64)
        np.cumsum(tri, out=base[1:])
        out = np.empty(max(int(base[-1]), 1), dtype=dtype)
        self._check(self._lib.lcsgpu_lcs_triangles_batch(self._ctx, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                         offs.ctypes.data_as(C.POINTER(C.c_int64)), len(groups),
                                                         out.ctypes.data, out.itemsize))
        return [out[base[g]:base[g + 1]].copy() for g in range(len(groups))]

    def lcs_triangle_dev(self, row_begin, row_end, d_out_ptr, elem_size, sync=False):
        self._check(self._lib.lcsgpu_lcs_triangle_dev(self._ctx, row_begin, row_end, C.c_void_p(d_out_ptr),
                                                      elem_size, 1 if sync else 0))

    def row_minima_dev(self, d_tri_ptr, elem_size, row_begin, row_end, kind, d_out_ptr, sync=False):
        self._check(self._lib.lcsgpu_row_minima_dev(self._ctx, C.c_void_p(d_tri_ptr), elem_size, row_begin,
                                                    row_end, kind, C.c_void_p(d_out_ptr), 1 if sync else 0))

    def mst_prim(self, kind=1):
        """Edges of Prim's MST in insertion order: structured array (from, to, dist)."""
        out = np.zeros(max(self.n - 1, 0), dtype=MST_EDGE)
        self._check(self._lib.lcsgpu_mst_prim(self._ctx, kind, out.ctypes.data if out.size else None))
        return out

    # ---- sharded MST (Boruvka over row blocks; one context per GPU) ----
    def mst_shard_begin(self, d_tri_ptr, elem_size, row_begin, row_end, kind=1):
        """kind may carry MST_TRIANGLE_ORIENTATION and MST_COMPUTE (d_tri_ptr is then an output, or None = no
        triangle is kept and every round recomputes the block)."""
        self._check(self._lib.lcsgpu_mst_shard_begin(self._ctx, C.c_void_p(d_tri_ptr) if d_tri_ptr else None, elem_size,
                                                     row_begin, row_end, kind))

    def mst_shard_best(self, d_keys_ptr=None, host=False):
        """Local half of a round into device memory (d_keys_ptr) and/or a host array (returned when host=True)."""
        h = np.zeros(self.n, dtype=MST_KEY) if host else None
        self._check(self._lib.lcsgpu_mst_shard_best(self._ctx, C.c_void_p(d_keys_ptr) if d_keys_ptr else None,
                                                    h.ctypes.data if host else None))
        return h

    def mst_shard_merge(self, d_gathered_ptr, n_parts):
        cnt = C.c_int32(0)
        self._check(self._lib.lcsgpu_mst_shard_merge(self._ctx, C.c_void_p(d_gathered_ptr) if d_gathered_ptr else None,
                                                     n_parts, C.byref(cnt)))
        return cnt.value

    def mst_shard_set_components(self, comp):
        comp = np.ascontiguousarray(comp, dtype=np.int32)
        assert len(comp) == self.n
        self._check(self._lib.lcsgpu_mst_shard_set_components(self._ctx, comp.ctypes.data))

    def mst_shard_finish(self, ordered=True):
        """The n-1 tree edges: in Prim's insertion order, or (ordered=False) as the rounds found them --
        mst_order_edges() then orders them, e.g. while the GPU already works on the next problem."""
        out = np.zeros(max(self.n - 1, 0), dtype=MST_EDGE)
        fn = self._lib.lcsgpu_mst_shard_finish if ordered else self._lib.lcsgpu_mst_shard_edges
        self._check(fn(self._ctx, out.ctypes.data if out.size else None))
        return out

    def upgma(self, kind=1, modified=False):
        """Children (left, right) of internal nodes n..2n-2 of the UPGMA tree."""
        left = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        right = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        self._check(self._lib.lcsgpu_upgma(self._ctx, kind, int(modified), left.ctypes.data, right.ctypes.data))
        return left[: max(self.n - 1, 0)], right[: max(self.n - 1, 0)]

    def nj(self, kind=1):
        """Children (left, right) of internal nodes n..2n-2 of the neighbour-joining tree."""
        left = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        right = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        self._check(self._lib.lcsgpu_nj(self._ctx, kind, left.ctypes.data, right.ctypes.data))
        return left[: max(self.n - 1, 0)], right[: max(self.n - 1, 0)]

    def assign_seeds(self, seed_ids, col_ids, dist, assign, first_k=1, kind=1):
        """Nearest-seed update of (dist, assign) over the columns, in place (float32 / int32 arrays)."""
        s_arr, s_ptr = _ids(seed_ids)
        c_arr, c_ptr = _ids(col_ids)
        assert dist.dtype == np.float32 and assign.dtype == np.int32 and len(dist) == len(c_arr) == len(assign)
        self._check(self._lib.lcsgpu_assign_seeds(self._ctx, s_ptr, len(s_arr), c_ptr, len(c_arr), kind, first_k,
                                                  dist.ctypes.data, assign.ctypes.data))

    def clarans(self, ids, n_medoids, n_fixed=1, explore_fraction=0.1, num_local=2, kind=1):
        """CLARANS medoids (member numbers within `ids`) of the sample `ids`, computed on the device."""
        arr, ptr = _ids(ids)
        out = np.zeros(max(n_medoids, 1), dtype=np.int32)
        self._check(self._lib.lcsgpu_clarans(self._ctx, ptr, len(arr), kind, n_medoids, n_fixed, explore_fraction,
                                             num_local, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n_medoids]

    def clarans_batch(self, samples, n_medoids, n_fixed=1, explore_fraction=0.1, num_local=2, kind=1):
        """lcsgpu_clarans_batch: `samples` = list of id arrays, `n_medoids` = one count or a list; returns a list of arrays."""
        ks = np.ascontiguousarray([n_medoids] * len(samples) if np.isscalar(n_medoids) else n_medoids, dtype=np.int32)
        off = np.zeros(len(samples) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in samples])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(s, np.int32) for s in samples]) if samples else np.zeros(0, np.int32), dtype=np.int32)
        out = np.zeros(max(int(ks.sum()), 1), dtype=np.int32)
        p32 = C.POINTER(C.c_int32)
        self._check(self._lib.lcsgpu_clarans_batch(self._ctx, ids.ctypes.data_as(p32), off.ctypes.data_as(C.POINTER(C.c_int64)), len(samples), kind,
                                                   ks.ctypes.data_as(p32), n_fixed, explore_fraction, num_local, out.ctypes.data_as(p32)))
        cuts = np.concatenate([[0], np.cumsum(ks)])
        return [out[cuts[i]:cuts[i + 1]].copy() for i in range(len(samples))]

    def assign_seeds_batch(self, seeds, cols, kind=1):
        """lcsgpu_assign_seeds_batch: lists of seed id arrays and column id arrays (one pair per evaluation); returns
        (dist, assign) lists."""
        p32, p64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        so = np.zeros(len(seeds) + 1, dtype=np.int64)
        so[1:] = np.cumsum([len(s) for s in seeds])
        co = np.zeros(len(cols) + 1, dtype=np.int64)
        co[1:] = np.cumsum([len(c) for c in cols])
        s_all = np.ascontiguousarray(np.concatenate([np.asarray(s, np.int32) for s in seeds]), dtype=np.int32)
        c_all = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int32) for c in cols]), dtype=np.int32)
        dist = np.zeros(len(c_all), dtype=np.float32)
        assign = np.zeros(len(c_all), dtype=np.int32)
        self._check(self._lib.lcsgpu_assign_seeds_batch(self._ctx, s_all.ctypes.data_as(p32), so.ctypes.data_as(p64), c_all.ctypes.data_as(p32),
                                                        co.ctypes.data_as(p64), len(seeds), kind, dist.ctypes.data, assign.ctypes.data))
        return ([dist[co[i]:co[i + 1]] for i in range(len(cols))], [assign[co[i]:co[i + 1]] for i in range(len(cols))])

    def dist_text(self, names, blocks, kind=1, square=False, pid=False, n_slots=2):
        """-dist_export rows as text made on the device (lcsgpu_dist_text_*): `names` without '>', `blocks` =
        [(row_begin, row_end), ...] in order; returns the concatenated bytes of the blocks."""
        raw = [x.encode("latin-1") if isinstance(x, str) else bytes(x) for x in names]
        off = np.zeros(len(raw) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in raw], dtype=np.uint64)
        flags = (1 if square else 0) | (2 if pid else 0)
        self._check(self._lib.lcsgpu_dist_text_begin(self._ctx, b"".join(raw), off.ctypes.data, kind, flags, n_slots))
        out = []
        try:
            pending = []
            todo = list(blocks)

            def collect():
                slot = pending.pop(0)
                ptr, nb = C.c_void_p(0), C.c_uint64(0)
                self._check(self._lib.lcsgpu_dist_text_wait(self._ctx, slot, C.byref(ptr), C.byref(nb)))
                out.append(C.string_at(ptr.value, nb.value) if nb.value else b"")
                return slot

            free = list(range(n_slots))
            for r0, r1 in todo:
                slot = free.pop(0) if free else collect()
                self._check(self._lib.lcsgpu_dist_text_submit(self._ctx, slot, int(r0), int(r1)))
                pending.append(slot)
            while pending:
                collect()
        finally:
            self._lib.lcsgpu_dist_text_end(self._ctx)
        return b"".join(out)

    def sync(self):
        self._check(self._lib.lcsgpu_sync(self._ctx))

    def last_kernel_ms(self):
        ms = C.c_double(0)
        n = C.c_int32(0)
        self._check(self._lib.lcsgpu_last_kernel_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @staticmethod
    def _spec(x):
        if isinstance(x, tuple):
            return None, None, int(x[0]), int(x[1])
        arr, ptr = _ids(x)
        return arr, ptr, 0, len(arr)


class LcsGpuGroup:
    """Several contexts (one per GPU; on a 1-GPU box several on the same device) holding the same set:
    the lcsgpu_multi_* calls of include/lcsgpu.h."""

    def __init__(self, devices):
        self.engs = [LcsGpu(d) for d in devices]
        self._lib = self.engs[0]._lib
        self._arr = (C.c_void_p * len(self.engs))(*[e._ctx for e in self.engs])
        self.n = 0

    def _check(self, rc):
        self.engs[0]._check(rc)

    def close(self):
        for e in self.engs:
            e.close()

    def upload_seqs(self, seqs):
        for e in self.engs:
            e.upload_seqs(seqs)
        self.n = self.engs[0].n

    def upload(self, codes, offsets):
        for e in self.engs:
            e.upload(codes, offsets)
        self.n = self.engs[0].n

    def upload_ordered(self, codes, offsets, order):
        for e in self.engs:
            e.upload_ordered(codes, offsets, order)
        self.n = self.engs[0].n

    def lcs_triangle(self, row_begin=0, row_end=None, dtype=np.uint16):
        row_end = self.n if row_end is None else row_end
        count = row_end * (row_end - 1) // 2 - row_begin * (row_begin - 1) // 2
        out = np.empty(max(count, 0), dtype=dtype)
        self._check(self._lib.lcsgpu_multi_lcs_triangle(self._arr, len(self.engs), row_begin, row_end,
                                                        out.ctypes.data if out.size else None, out.itemsize))
        return out

    def mst_prim(self, kind=1):
        out = np.zeros(max(self.n - 1, 0), dtype=MST_EDGE)
        self._check(self._lib.lcsgpu_multi_mst_prim(self._arr, len(self.engs), kind, out.ctypes.data if out.size else None))
        return out

    def transport(self):
        """lcsgpu_multi_transport: how the contexts reach each other + the key exchange of the last mst_prim (text)."""
        buf = C.create_string_buffer(2048)
        self._check(self._lib.lcsgpu_multi_transport(self._arr, len(self.engs), buf, len(buf)))
        return buf.value.decode()

    def last_kernel_ms(self):
        """Per context: LCS kernel milliseconds of its row block in the last multi-context call."""
        return [e.last_kernel_ms()[0] for e in self.engs]

    def upgma(self, kind=1, modified=False):
        left = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        right = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        self._check(self._lib.lcsgpu_multi_upgma(self._arr, len(self.engs), kind, int(modified), left.ctypes.data, right.ctypes.data))
        return left[: max(self.n - 1, 0)], right[: max(self.n - 1, 0)]

    def nj(self, kind=1):
        left = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        right = np.zeros(max(self.n - 1, 1), dtype=np.int32)
        self._check(self._lib.lcsgpu_multi_nj(self._arr, len(self.engs), kind, left.ctypes.data, right.ctypes.data))
        return left[: max(self.n - 1, 0)], right[: max(self.n - 1, 0)]
