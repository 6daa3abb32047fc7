/*
 * lcsgpu.h -- C-ABI of the MI355X (gfx950) all-pairs bit-parallel LCS engine.
 *
 * This is the drop-in boundary for FAMSA's pairwise-similarity hot path.  FAMSA has
 * no FFI/plugin layer of its own; the seam is C++-internal.  Each
 * entry point below names the reference interface it replaces (paths relative to the
 * reference's src/).  A maintainer would bind these from the batch-distance templates
 * of tree/AbstractTreeGenerator.hpp -- INTEGRATION.md shows the binding.
 *
 * NO GPU, NO RESULT: this library is the gfx950 path and nothing else.  Without a usable device lcsgpu_create answers
 * LCSGPU_E_NODEVICE, and a HIP error inside any call comes back as LCSGPU_E_HIP -- there is no CPU implementation behind
 * this ABI and none is substituted silently (SURVEY.md 8b sketched one as "plumbing"; the build has none: a caller that
 * wants a CPU path keeps the reference's own CLCSBP).  The only calls that work without a device are the pure host
 * helpers: lcsgpu_version, lcsgpu_last_error, lcsgpu_device_count, lcsgpu_encode, lcsgpu_mst_merge_host,
 * lcsgpu_mst_order_edges.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * LCSGPU_E_* code, never throws; lcsgpu_last_error() gives a thread-local message.
 * The caller owns every buffer it passes.  A context is bound to one GPU and may be used from
 * several host threads at once (the reference runs one CLCSBP per worker thread): host-memory
 * calls (lcsgpu_lcs_rect, lcsgpu_lcs_triangle, lcsgpu_lcs_triangle_ids) execute concurrently on
 * internal lanes (HIP streams); device-memory calls, the tree reducers and lcsgpu_sync use one
 * fixed stream (lcsgpu_stream) in call order; lcsgpu_upload waits for everything and is
 * exclusive.  Different contexts are independent -- the multi-GPU model is one process (or one
 * context) per GPU; the lcsgpu_multi_* calls and the lcsgpu_mst_shard_* protocol let several contexts work
 * on one problem.
 *
 * Symbol codes are the reference's: index in "ARNDCQEGHILKMFPSTWYVBZX*"
 * (core/sequence.cpp:17), 22 = unknown; only codes < 20 ever match
 * (core/sequence.cpp:199).  All LCS values are ORIENTED: "ref" is the bit-mask side
 * (CSequence::p_bit_masks), "partner" is streamed, exactly as in
 * CLCSBP::GetLCSBP(ref, partners...) (lcs/lcsbp.h:36-46); the result reproduces the
 * reference's carry rule (lcs/lcsbp_classic.h:51-58) bit for bit, so LCS(ref=a,
 * partner=b) may differ from LCS(ref=b, partner=a) for sequences holding a 64-residue
 * aligned homopolymer word (SURVEY.md note Q).
 */
#ifndef LCSGPU_H
#define LCSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
typedef struct lcsgpu_ctx lcsgpu_ctx; /* (opaque; declared before the push below: the type itself exports nothing) */
/* the library is built with -fvisibility=hidden: the functions declared here are the functions it exports */
#pragma GCC visibility push(default)

#define LCSGPU_OK 0
#define LCSGPU_E_INVALID (-1)  /* bad argument */
#define LCSGPU_E_NODEVICE (-2) /* no usable gfx950 device / HIP runtime error at create */
#define LCSGPU_E_HIP (-3)      /* a HIP call failed; see lcsgpu_last_error() */
#define LCSGPU_E_NOMEM (-4)
#define LCSGPU_E_STATE (-5)    /* e.g. compute before upload */
#define LCSGPU_E_UNSUPPORTED (-6) /* shape outside what a device reducer handles; the host form applies */

/* Library / build information: "lcsgpu <version> gfx950 recolor=<on|off|failed> kernels=<id>/<id>" -- kernels= names the
 * device code of the two LCS translation units (sha256 prefix of the listings they were assembled from, without the
 * __hip_cuid_<hash> symbol the compiler derives from the output path: equal ids = the same kernels instruction for
 * instruction wherever they were built; bench.py matches committed profiles to the running library by it);
 * recolor= says how the hot kernels were built: with the register-bank renaming pass and its equivalence check passed ("on"), without the pass
 * on request ("off", make RECOLOR=0), or as compiled because the pass or its check failed ("failed": results are the
 * same, the LCS kernels ~5 % slower). */
const char* lcsgpu_version(void);
const char* lcsgpu_last_error(void);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int lcsgpu_device_count(void);

/* Create / destroy an engine bound to HIP device `device_id`.
 * Replaces: CLCSBP::CLCSBP(instruction_set_t) (lcs/lcsbp.cpp:24-45), one per worker. */
int lcsgpu_create(int device_id, lcsgpu_ctx** out_ctx);
int lcsgpu_destroy(lcsgpu_ctx* ctx);
/* Optional: a context creates the stream pairs of its internal lanes when a call first needs them (~20 ms each;
 * a single-threaded caller never needs a second one) and hands out at most 16 of them at a time.  A caller that is
 * about to issue host-memory calls from `n_threads` threads (the FastTree recursion) raises that limit to n_threads
 * (at most 64; lanes beyond 16 share the 16 hardware queues) and has the first 16 created now -- e.g. while it still
 * reads its input. */
int lcsgpu_reserve_lanes(lcsgpu_ctx* ctx, int32_t n_threads);

/* Residue characters -> symbol codes, gaps ('-') dropped.
 * Replaces: the encoding loop of CSequence::CSequence (core/sequence.cpp:53-79).
 * `codes` must hold n bytes; *n_codes receives the count written.  Host only. */
int lcsgpu_encode(const char* residues, size_t n, uint8_t* codes, size_t* n_codes);

/* Upload a sequence set (replaces any previous one): codes[offsets[i] .. offsets[i+1])
 * is sequence i, unpadded, n sequences, ids 0..n-1 in the caller's order.
 * Replaces: the per-sequence state the reference prepares on the host --
 * CSequence::data/length (core/sequence.h:28-32), the padding of
 * CFAMSA::sortAndExtendSequences/extendSequences (msa.cpp:245-317; the engine pads
 * internally) and CSequence::ComputeBitMasks (core/sequence.cpp:190-201; masks are
 * rebuilt on the device per launch). */
int lcsgpu_upload(lcsgpu_ctx* ctx, const uint8_t* codes, const uint64_t* offsets, int32_t n);

/* The same for a caller that holds its records in another order than it wants the set in: sequence k of the
 * uploaded set (k < n) is record order[k] of (codes, offsets), 0 <= order[k] < n_records; records that no entry
 * names are left out, a record may be named more than once.  order == NULL: lcsgpu_upload (n == n_records).
 * Replaces: the re-ordering of the sequence vector itself in CFAMSA::sortAndExtendSequences (msa.cpp:245-279:
 * stable_sort of the CSequence objects) and the erase of CFAMSA::removeDuplicates (msa.cpp:338-356) -- the caller
 * computes the order and which records are kept (ids 0..n-1 = ranks in that order, msa.cpp:559-561), the residues
 * are gathered on the device instead of being packed on the host first.
 * The WHOLE record buffer codes[0 .. offsets[n_records]) and all n_records + 1 offsets are staged on the device for the
 * gather, whatever the order selects (transient device memory and host-to-device time are those of the records, not of
 * the set); offsets must be non-decreasing from offsets[0] == 0 over the records the order names, and
 * offsets[n_records] must be the end of the buffer. */
int lcsgpu_upload_ordered(lcsgpu_ctx* ctx, const uint8_t* codes, const uint64_t* offsets, int32_t n_records,
                          const int32_t* order, int32_t n);

/* Number of sequences / length of sequence i currently uploaded (negative on error). */
int32_t lcsgpu_count(lcsgpu_ctx* ctx);
int32_t lcsgpu_length(lcsgpu_ctx* ctx, int32_t i);

/* Which uploaded sequences are orientation sensitive AS A REF, i.e. hold an aligned 64-residue
 * homopolymer word at word index >= 1, the only case in which the reference's carry rule
 * (lcs/lcsbp_classic.h:55-56) departs from a true LCS.  flags (may be NULL) receives n bytes of
 * 0/1; the return value is the number of such sequences (>= 0) or a negative error code.  When it
 * is 0, LCS(ref=a, partner=b) == LCS(ref=b, partner=a) for every pair of the set, so a consumer
 * that needs both orientations (MSTPrim: ref = the node just added) can use one triangle. */
int32_t lcsgpu_orientation_flags(lcsgpu_ctx* ctx, uint8_t* flags);

/* Rectangle of oriented LCS lengths into HOST memory:
 *   out[r * ld + c] = LCS(ref = ref_ids[r], partner = col(c)),  r < n_refs, c < n_cols
 * col(c) = col_ids[c] if col_ids != NULL, else col_begin + c.  elem_size is 2 (uint16_t)
 * or 4 (uint32_t); 2 is rejected if any uploaded sequence is longer than 65535.
 * ref_ids == NULL means refs ref_begin .. ref_begin+n_refs-1.
 * Replaces: calculateDistanceVector (tree/AbstractTreeGenerator.hpp:130-182: one ref x a
 * contiguous partner array) and calculateDistanceRange / calculateDistanceRangeSV
 * (hpp:190-375: one ref x an id list), i.e. the CLCSBP::GetLCSBP groups of 8
 * (lcs/lcsbp.cpp:163,267) for many refs at once, without the Transform. */
int lcsgpu_lcs_rect(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                    const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* out,
                    int64_t ld, int elem_size);

/* Same, result left in DEVICE memory (d_out is a device pointer on ctx's GPU, e.g. a
 * torch tensor's data_ptr()); asynchronous on the context's stream unless `sync` != 0.
 * ref_ids / col_ids are HOST arrays (copied by the call). */
int lcsgpu_lcs_rect_dev(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                        const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* d_out,
                        int64_t ld, int elem_size, int sync);

/* Rows [row_begin, row_end) of the lower triangle, ref = row i, partner = column j < i:
 *   out[i*(i-1)/2 + j - row_begin*(row_begin-1)/2] = LCS(ref = i, partner = j)
 * i.e. TriangleMatrix::access(i, j) (tree/TreeDefs.h:115-120) relative to the first row.
 * Rows are independent, so a node's GPUs take disjoint row blocks (no data-path
 * collective).  Host-memory and device-memory forms.
 * Replaces: calculateDistanceMatrix (tree/AbstractTreeGenerator.hpp:378-398) and the row
 * producers UPGMA::computeDistances (tree/UPGMA.cpp:75-109), SingleLinkage::run workers
 * (tree/SingleLinkage.cpp:48-82), DistanceCalculator::run workers
 * (tree/DistanceCalculator.cpp:28-82). */
int lcsgpu_lcs_triangle(lcsgpu_ctx* ctx, int32_t row_begin, int32_t row_end, void* out,
                        int elem_size);
int lcsgpu_lcs_triangle_dev(lcsgpu_ctx* ctx, int32_t row_begin, int32_t row_end, void* d_out,
                            int elem_size, int sync);

/* Lower triangle over an ID LIST (host memory): out[k*(k-1)/2 + c] = LCS(ref = ids[k], partner = ids[c]),
 * c < k < n_ids -- calculateDistanceMatrix (tree/AbstractTreeGenerator.hpp:378-398) on a subset of the
 * uploaded sequences: the sample matrix of FastTree::clusterSeeds (tree/FastTree.cpp:415) and the
 * per-cluster matrices of the partial generators (UPGMA::runPartial, tree/UPGMA.cpp:55-70). */
int lcsgpu_lcs_triangle_ids(lcsgpu_ctx* ctx, const int32_t* ids, int32_t n_ids, void* out, int elem_size);

/* Distance measures of the reference's Transform functors (tree/AbstractTreeGenerator.hpp:28-82,
 * enum Distance in tree/TreeDefs.h:20-27). */
#define LCSGPU_DIST_INDEL_DIV_LCS 0    /* (double)indel / lcs                           */
#define LCSGPU_DIST_INDEL075_DIV_LCS 1 /* pow(indel, 0.75) / lcs  (the CLI default)      */

/* One row's nearest neighbour among the columns j < i of the lower triangle. */
typedef struct lcsgpu_rowmin {
    double dist;   /* Transform<double, kind>(lcs, len_i, len_j); DBL_MAX if the row is empty */
    int64_t index; /* the column j attaining it; -1 if the row is empty                       */
} lcsgpu_rowmin;

/* Per-row minima over a DEVICE-resident triangle slice produced by lcsgpu_lcs_triangle_dev for
 * the same [row_begin, row_end):  out[i - row_begin] = min over j < i of the key
 *   (d(i,j), ~((uint64)j << 32 | i))   compared lexicographically,
 * which is the edge order of MSTPrim (tree/MSTPrim.cpp:493-509, mst_edge_t / ids_to_uint64 in
 * tree/MSTPrim.h:432-483): smaller distance first, and among equal distances the LARGER j.
 * d is computed in double exactly as Transform<double, kind> does: indel = len_i + len_j - 2*lcs,
 * pow(indel, 0.75) taken from a table the library fills on the HOST with libm's pow (so the
 * values are the reference's), IEEE double division on the device; lcs == 0 gives
 * nextafter(DBL_MAX, 0).  This is the payload of the single-linkage exchange step: each GPU
 * reduces its own row block, then the ranks allgather n x 16 bytes (RCCL over xGMI).
 * d_out is a device pointer to (row_end - row_begin) lcsgpu_rowmin records. */
int lcsgpu_row_minima_dev(lcsgpu_ctx* ctx, const void* d_triangle, int elem_size, int32_t row_begin,
                          int32_t row_end, int distance_kind, void* d_out, int sync);

/* One edge of the minimum spanning tree, in the order Prim's algorithm adds them. */
typedef struct lcsgpu_mst_edge {
    int32_t from, to; /* from < to */
    double dist;      /* Transform<double, kind> of the pair, ref = the endpoint that was in the tree */
} lcsgpu_mst_edge;

/* Prim's MST over the complete graph of the uploaded set: the LCS triangle is computed into HBM and
 * the tree is built there -- by Boruvka rounds over the triangle when distances do not depend on the
 * ref / partner roles (the edge order below is strict and total, so the tree is unique whatever the
 * algorithm), else by n-1 relaxation steps (one launch per step).
 * out_edges (HOST, n-1 records) receives the edges in the order Prim adds them, starting from
 * vertex 0.  Semantics are those of MSTPrim::run_view (tree/MSTPrim.cpp:280-549): distance
 * d(cur, v) uses LCS(ref = cur, partner = v); keys are (d, ~((uint64)min(cur,v) << 32 | max(cur,v)))
 * compared lexicographically (mst_edge_t / dist_t, tree/MSTPrim.h:424-483), so the tree and the
 * order are unique and identical to the reference's.  What stays on the host is the MST ->
 * dendrogram conversion (tree/MSTPrim.cpp:784-833), which is O(n log n).
 * Replaces: the per-step distance batches (calculateDistanceRangeSV, hpp:287-375), the key update
 * and the candidate selection of MSTPrim::run_view. */
int lcsgpu_mst_prim(lcsgpu_ctx* ctx, int distance_kind, lcsgpu_mst_edge* out_edges);
/* OR this into distance_kind to take every distance from LCS(ref = larger id, partner = smaller id),
 * the orientation SingleLinkage (SLINK) sees (tree/SingleLinkage.cpp:58-81), instead of MSTPrim's
 * LCS(ref = the node just added).  The two differ only for orientation-sensitive sequences. */
#define LCSGPU_MST_TRIANGLE_ORIENTATION 0x100

/* ---- Single linkage over SEVERAL GPUs: the N x N pair space tiled by row block (one context / process per
 * GPU, each holding the whole uploaded set and rows [row_begin, row_end) of the LCS triangle in its HBM),
 * Boruvka rounds with ONE exchange per round.  The edge order of MSTPrim (tree/MSTPrim.cpp:493-509,
 * tree/MSTPrim.h:424-483) is strict and total, so the minimum spanning tree is unique and this yields
 * exactly the edges MSTPrim::run_view (tree/MSTPrim.cpp:356-533) finds; their insertion order from vertex 0
 * (cpp:372-391), which the dendrogram needs (cpp:784-833), is a walk over the tree (lcsgpu_mst_order_edges).
 *
 * A round:  (1) every context: lcsgpu_mst_shard_best -> n keys of 16 bytes: per vertex, the best edge into
 * another component among the pairs of the context's row block (round 0: the per-row minima, completed by
 * the per-column minima);  (2) the caller exchanges the key arrays -- an all-gather of n x 16 B per GPU
 * (RCCL over xGMI between processes: torch.distributed.all_gather_into_tensor in bench.py; host memory
 * between the contexts of one process);  (3) every context: lcsgpu_mst_shard_merge over the n_parts x n
 * gathered keys (device memory), or one lcsgpu_mst_merge_host over them in host memory followed by
 * lcsgpu_mst_shard_set_components on every context.  Rounds repeat until n-1 edges are found (<= log2 n).
 * The component labels are a pure function of the exchanged keys: every context derives the same ones.
 *
 * Not available (LCSGPU_E_UNSUPPORTED) when an uploaded sequence is orientation sensitive and MSTPrim's own
 * orientation is asked for (no LCSGPU_MST_TRIANGLE_ORIENTATION): lcsgpu_mst_prim handles that on one GPU. */
typedef struct lcsgpu_mst_key {
    uint64_t dist_bits; /* bit pattern of the double distance (>= 0, so the bits order like the values); bits of DBL_MAX = none */
    uint64_t id;        /* ~(((uint64_t)min(u,v) << 32) + max(u,v));  ~0 = none */
} lcsgpu_mst_key;

/* d_triangle: DEVICE memory holding rows [row_begin, row_end) as lcsgpu_lcs_triangle_dev wrote them; it must
 * stay valid until the last lcsgpu_mst_shard_best.  Resets the component state (every vertex on its own).
 * distance_kind may carry LCSGPU_MST_TRIANGLE_ORIENTATION and LCSGPU_MST_COMPUTE.
 *
 * LCSGPU_MST_COMPUTE -- the local half of a round FUSED into the LCS launch: the workgroup that holds LCS(row, 256
 * columns) in registers folds it into every vertex's best edge into another component before (or instead of)
 * storing it -- an integer pre-filter on the LCS length, the exact key (Transform<double> + MSTPrim's id order) for
 * survivors only, per-vertex 64-bit records updated by compare-and-swap -- so nothing but n x 16 B of keys leaves
 * the launch: "each GPU reduces its row block to per-row minima, then the ranks all-gather them".
 *   d_triangle != NULL: it is an OUTPUT (uint16, elem_size 2): this call computes rows [row_begin, row_end) into it as
 *     lcsgpu_lcs_triangle_dev would (lcsgpu_last_kernel_ms times that launch) AND round 0's local half in the same
 *     launch; later rounds read the resident triangle.
 *   d_triangle == NULL: NO triangle is kept: every lcsgpu_mst_shard_best recomputes the block's LCS values with the
 *     fold fused in.  O(n) device memory like the reference's MSTPrim (tree/MSTPrim.cpp:450-512), at the price of
 *     one LCS pass per round (<= log2 n rounds; 7 at n = 100 000) -- what lifts the n <= ~530 000 limit of a
 *     2-byte-per-pair triangle in 288 GB.
 * Needs every sequence <= 65535 residues (LCSGPU_E_UNSUPPORTED otherwise). */
#define LCSGPU_MST_COMPUTE 0x200
int lcsgpu_mst_shard_begin(lcsgpu_ctx* ctx, void* d_triangle, int elem_size, int32_t row_begin, int32_t row_end,
                           int distance_kind);
/* Local half of a round.  d_keys: DEVICE buffer of n lcsgpu_mst_key (NULL = a buffer of the context);
 * h_keys: if not NULL, the keys are also copied to this HOST buffer and the call returns when they are
 * there; otherwise the work is queued on the context's stream (lcsgpu_stream) and the call returns at once. */
int lcsgpu_mst_shard_best(lcsgpu_ctx* ctx, void* d_keys, lcsgpu_mst_key* h_keys);
/* Global half of a round on the device: d_gathered = n_parts x n keys (DEVICE; part p at d_gathered + p*n),
 * every part's lcsgpu_mst_shard_best output of this round.  *n_edges (may be NULL) = tree edges found so far
 * (the call waits for the round). */
int lcsgpu_mst_shard_merge(lcsgpu_ctx* ctx, const void* d_gathered, int32_t n_parts, int32_t* n_edges);
/* After the last round (n-1 edges): the edges in the order Prim's algorithm adds them from vertex 0, as
 * lcsgpu_mst_prim returns them (HOST, n-1 records). */
int lcsgpu_mst_shard_finish(lcsgpu_ctx* ctx, lcsgpu_mst_edge* out_edges);
/* The same n-1 edges in the order the rounds found them (any order): for a caller that overlaps the ordering
 * (lcsgpu_mst_order_edges, O(n log n) on the host, ~10 ms at n = 100 000) with the GPU work of its next problem. */
int lcsgpu_mst_shard_edges(lcsgpu_ctx* ctx, lcsgpu_mst_edge* out_edges);
/* Host form of the global half -- no GPU, no context: keys = n_parts x n (HOST); comp[n] (in/out) the
 * component label of every vertex (initially comp[v] = v); the round's new tree edges are appended to
 * edges[*n_edges ...] and *n_edges is advanced.  Afterwards hand comp to every context: */
int lcsgpu_mst_merge_host(const lcsgpu_mst_key* keys, int32_t n_parts, int32_t n, int32_t* comp, lcsgpu_mst_edge* edges,
                          int32_t* n_edges);
int lcsgpu_mst_shard_set_components(lcsgpu_ctx* ctx, const int32_t* comp);
/* The n-1 edges of a spanning tree, in place, into Prim's insertion order from vertex 0 with MSTPrim's
 * candidate order (tree/MSTPrim.cpp:372-391).  Host only. */
int lcsgpu_mst_order_edges(lcsgpu_mst_edge* edges, int32_t n);

/* UPGMA (or MAFFT-style "modified" UPGMA) over the uploaded set, entirely on the device: LCS
 * triangle -> float distances (Transform<float, kind>: host-built (float)pow(indel,0.75) table,
 * IEEE float division) -> n-1 merges with the nearest-neighbour-array algorithm of the reference,
 * operation for operation (strict '<' scans, stale row minima, (x+y)*0.5f resp.
 * 0.05f*(x+y)+0.9f*min(x,y) without contraction), so the result equals UPGMA::computeTree's.
 * out_left/out_right (HOST, n-1 entries each): children of internal node n+k, k = 0..n-2, ids as in
 * tree_structure (leaves 0..n-1).  Returns LCSGPU_E_INVALID for inputs on which the reference's
 * algorithm is undefined (no finite nearest neighbour, e.g. a sequence with LCS 0 to all others).
 * How the n-1 merges run (all forms give the reference's tree bit for bit): on an n x (n + n/10) float matrix
 * -- live clusters keep a row, new clusters take the next free column, the columns are compacted in place when they run
 * out; 46 GB at 100 000 sequences, the LCS triangle itself only ever exists block-wise -- in batches of up to 32 merges per
 * three launches: the next picks are the next entries of the rows' sorted (min_dist, index) order, checked afterwards
 * against the keys of the clusters the batch created (LCSGPU_UPGMA_BATCH=0|8|16|32).  Where that matrix does not fit: the
 * n x n matrix, then the packed float triangle, with one launch per merge (LCSGPU_UPGMA_LAYOUT=square|triangle forces one).
 * Replaces: UPGMA::computeDistances + UPGMA::computeTree (tree/UPGMA.cpp:75-109, 114-295). */
int lcsgpu_upgma(lcsgpu_ctx* ctx, int distance_kind, int modified, int32_t* out_left, int32_t* out_right);

/* Neighbour joining over the uploaded set on the device: LCS triangle -> float distances ->
 * n-2 merges, each the reference's (tree/NeighborJoining.cpp:33-118) float arithmetic in its
 * exact association and summation order, first strict minimum of q over i < j.  Output as for
 * lcsgpu_upgma (children of internal nodes n..2n-2).
 * Replaces: calculateDistanceMatrix (single-threaded in the reference, NeighborJoining.cpp:16) +
 * NeighborJoining::computeTree. */
int lcsgpu_nj(lcsgpu_ctx* ctx, int distance_kind, int32_t* out_left, int32_t* out_right);

/* ---- Several contexts (one per GPU of a node) on ONE problem, driven from one host thread --------------
 * SURVEY 8(b) sketched lcsgpu_create(dev_ids, n_dev); the boundary keeps one context per GPU (the reference's
 * one-CLCSBP-per-worker convention) and adds calls that take a LIST of contexts.  Every context of the list
 * must hold the same uploaded set (lcsgpu_upload on each); contexts may also share a device.  The pair space
 * is tiled by row blocks of equal pair counts, block k on ctxs[k]; all GPUs compute at the same time.
 *
 * lcsgpu_multi_lcs_triangle: as lcsgpu_lcs_triangle, HOST output -- the row producers of UPGMA::computeDistances
 *   (tree/UPGMA.cpp:75-109) / DistanceCalculator::run (tree/DistanceCalculator.cpp:28-82), one block per GPU; the
 *   blocks come back over all the GPUs' PCIe links at the same time (one host thread per context drains its block).
 * lcsgpu_multi_upgma / lcsgpu_multi_nj: as lcsgpu_upgma / lcsgpu_nj; the row blocks of the other GPUs are copied
 *   into ctxs[0]'s HBM over xGMI (device-to-device), where the sequential merges run.
 * lcsgpu_multi_mst_prim: as lcsgpu_mst_prim, by the sharded Boruvka rounds above with the exchange in DEVICE memory, in one
 *   of two forms.  (a) RCCL: ONE grouped ncclAllGather of n keys (16 B each) per context and round, in place, on the
 *   contexts' streams (librccl is loaded on demand and a communicator set per device list is kept for the life of the
 *   process) -- the "RCCL allgather of per-row minima over xGMI".  (b) Peer copies: every context pushes its n keys into
 *   its slot of every other context's gathered buffer (N x (N-1) copies on the producers' streams).  Either way every
 *   context runs the same global half on its own GPU and the host synchronises once per round (the edge count).
 *   LCSGPU_EXCHANGE in the environment chooses: unset / "auto" = (a) when there are >= 2 contexts, each on its own
 *   device, and RCCL loads and initialises, else (b) with the reason kept for lcsgpu_multi_transport; "rccl" = (a) or an
 *   error -- LCSGPU_E_UNSUPPORTED when two contexts share a device (RCCL wants one device per rank) or librccl is
 *   missing; "peer" = (b).  The first round a communicator set ever serves is cross-checked: the round's keys also
 *   travel by (b) into the idle half of the gathered buffers and the two results are compared byte for byte on the host
 *   (LCSGPU_E_HIP naming the context pair on a difference; LCSGPU_EXCHANGE_CHECK=0 skips it, =always repeats it every
 *   call).  Before the edge list of context 0 is returned every other context is synchronised and must report no HIP
 *   error, no inconsistent keys and the same edge count.  LCSGPU_E_UNSUPPORTED also for orientation-sensitive sets in
 *   MSTPrim's own orientation (run lcsgpu_mst_prim on one context then).  lcsgpu_last_kernel_ms afterwards answers for
 *   EVERY context of the call (its own block's LCS launch), so a caller sees how balanced the row blocks ran.
 * Device-to-device copies between contexts switch peer access on for the device pair at first use; where the
 * devices cannot address each other the copy is staged through pinned host memory (slower, same result).  Test
 * switch (environment): LCSGPU_TRANSPORT=peer makes same-device contexts take the hipMemcpyPeerAsync branch too (so a
 * one-GPU box runs it), LCSGPU_TRANSPORT=host makes every copy between contexts take the pinned-host path. */
int lcsgpu_multi_lcs_triangle(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int32_t row_begin, int32_t row_end, void* out,
                              int elem_size);
int lcsgpu_multi_upgma(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, int modified, int32_t* out_left,
                       int32_t* out_right);
int lcsgpu_multi_nj(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, int32_t* out_left, int32_t* out_right);
int lcsgpu_multi_mst_prim(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, lcsgpu_mst_edge* out_edges);
/* How the contexts of a list reach each other, as text for a log (lines separated by '\n', NUL terminated, truncated to
 * cap bytes): the transport a copy between every ordered pair of contexts takes -- "peer-copy" (hipMemcpyPeerAsync with
 * peer access on: one xGMI hop), "host-staging" (no peer access: pinned host memory, two PCIe crossings),
 * "same-device" -- counted per kind, the pairs that get no peer copy named; and which key exchange the last
 * lcsgpu_multi_mst_prim of this process used and why (see above).  Asking switches peer access on for the pairs, as
 * the first copy would.  famsa-gpu -v prints it.  Nothing comparable exists in the reference (one process, host
 * threads: tree/MSTPrim.cpp:330-538). */
int lcsgpu_multi_transport(lcsgpu_ctx* const* ctxs, int32_t n_ctx, char* buf, size_t cap);

/* The lower triangles of several id lists in one call (the leaf sub-trees of one FastTree split):
 * list g = ids[group_offsets[g] .. group_offsets[g+1]), m_g members; out receives the packed
 * triangles one after the other, list g at element offset sum_{h<g} m_h(m_h-1)/2, inside it
 * out[k*(k-1)/2 + c] = LCS(ref = list[k], partner = list[c]), c < k  (as lcsgpu_lcs_triangle_ids).
 * out is HOST memory.  One kernel launch per word-count class for the whole batch instead of
 * one call per list.
 * Replaces: the calculateDistanceMatrix calls of the leaf generators that FastTree::doStep runs
 * one by one on its sub-trees (tree/FastTree.cpp:82-103, 180-246). */
int lcsgpu_lcs_triangles_batch(lcsgpu_ctx* ctx, const int32_t* ids, const int64_t* group_offsets, int32_t n_groups,
                               void* out, int elem_size);

/* Seed assignment of one FastTree evaluation: for r = 0 .. n_seeds-1 in order,
 *   d = Transform<float>(LCS(ref = seed_ids[r], partner = col_ids[j]));  if (d < dist[j]) { dist[j] = d; assign[j] = first_k + r; }
 * dist / assign (HOST, n_cols entries each) are read and updated: the caller initialises them with the
 * distances to its seed 0 and zeros.  The LCS rectangle stays in HBM; 8 bytes per column come back.
 * Replaces: the seeds x all calculateDistanceVector sweep of FastTree::makeEvaluation
 * (tree/FastTree.cpp:309-324). */
int lcsgpu_assign_seeds(lcsgpu_ctx* ctx, const int32_t* seed_ids, int32_t n_seeds, const int32_t* col_ids,
                        int32_t n_cols, int distance_kind, int32_t first_k, float* dist, int32_t* assign);

/* CLARANS k-medoids over a sample of the uploaded set, on the device: LCS triangle over `ids`
 * (ref = ids[i], partner = ids[j], j < i) -> float distances (Transform<float>) -> `num_local`
 * local searches, each the reference's swap search with its two mt19937 streams, its float
 * operation order, comparison directions and tie rules, so the medoids are the reference's.
 * ids (HOST, n_ids entries): the sample in the caller's order (the order defines the member
 * numbers the result refers to).  medoids_out (HOST, n_medoids entries): member numbers 0..n_ids-1
 * of the chosen medoids, slot order; slots < n_fixed are never swapped (the reference pins member 0).
 * Returns LCSGPU_E_UNSUPPORTED when the shape is outside what the device search handles
 * (n_medoids > 1024 or n_ids - n_medoids > 2048): the caller then runs its own host search.
 * Thread-safe (each call runs its own launches; lcsgpu_clarans_batch is the form for many samples).
 * Replaces: the sample matrix + CLARANS::operator() in FastTree::clusterSeeds
 * (tree/FastTree.cpp:412-417, tree/Clustering.cpp:17-305). */
int lcsgpu_clarans(lcsgpu_ctx* ctx, const int32_t* ids, int32_t n_ids, int distance_kind, int32_t n_medoids,
                   int32_t n_fixed, float explore_fraction, int32_t num_local, int32_t* medoids_out);

/* ---- -dist_export: the rows of the distance matrix as TEXT, made on the device ------------------------------------
 * Replaces: the row workers AND the single writer thread of DistanceCalculator::run (tree/DistanceCalculator.cpp:28-113):
 * calculateDistanceVector per row, the float row vector, "<id>," + num2str over the row with
 * NumericConversions::Double2PChar(v, 6) (utils/conversion.h:109-119) and the ',' -> '\n' at the row's end.  A row block
 * goes LCS rectangle -> final bytes without leaving HBM as numbers; the caller receives the block's text -- rows
 * row_begin .. row_end-1 one after the other, byte for byte what the reference writes for them -- in pinned host memory
 * owned by the context and only has to put it into its file.  Rows are those of the set AS UPLOADED (the reference
 * exports before it sorts: msa.cpp, DistanceCalculator runs on the input order); row i holds the values j < i, or all n
 * with LCSGPU_TEXT_SQUARE (the header line of -square_matrix is the caller's: it needs no LCS).
 *   begin : ids = the sequences' names without the leading '>', name i = ids[id_offsets[i] .. id_offsets[i+1]) (HOST;
 *           copied); distance_kind as above, ignored with LCSGPU_TEXT_PID (Transform<float, pairwise_identity>, -pid);
 *           n_slots (1..8) = how many blocks the caller wants to have in flight.  Replaces any earlier begin.
 *   submit: queue rows [row_begin, row_end) (at most 32768) on a free slot; returns at once.  All slots of a context
 *           compute on its stream in submit order; a finished block's text travels while the next one is computed.
 *   wait  : block until the slot's text is in host memory; *text stays valid until the slot's next submit (or end).
 *   end   : frees the slots (an upload or lcsgpu_destroy does so too).
 * The slots of one context are driven from ONE host thread; different contexts are independent, so with several GPUs
 * the caller deals its blocks round robin (host/trees.cpp).  Device memory per slot is sized for the longest value a
 * row could hold (39 bytes, the lcs == 0 form), host memory for the block as it came out. */
#define LCSGPU_TEXT_SQUARE 0x1
#define LCSGPU_TEXT_PID 0x2
int lcsgpu_dist_text_begin(lcsgpu_ctx* ctx, const char* ids, const uint64_t* id_offsets, int distance_kind, int flags,
                           int32_t n_slots);
int lcsgpu_dist_text_submit(lcsgpu_ctx* ctx, int32_t slot, int32_t row_begin, int32_t row_end);
int lcsgpu_dist_text_wait(lcsgpu_ctx* ctx, int32_t slot, const char** text, uint64_t* n_bytes);
int lcsgpu_dist_text_end(lcsgpu_ctx* ctx);

/* ---- The FastTree recursion level by level: all splits of a level in one call ------------------------------------
 * lcsgpu_clarans_batch: lcsgpu_clarans for n_jobs samples at once -- sample g = ids[offsets[g] .. offsets[g+1]) with
 *   n_medoids[g] medoids (slots < n_fixed pinned); medoids_out receives the samples' medoids one after the other.  One
 *   batched LCS launch computes every sample's triangle, one launch their float matrices, and ONE workgroup per sample
 *   runs that sample's whole chain of num_local local searches (shuffle, start, search, keep the cheaper one) without
 *   the host in between; up to 512 chains are resident at once and nothing else competes for the CUs meanwhile.
 *   LCSGPU_E_UNSUPPORTED (nothing computed) if any sample is outside the device search's shapes.
 *   Replaces: FastTree::clusterSeeds' sample matrix + CLARANS::operator() (tree/FastTree.cpp:412-417,
 *   tree/Clustering.cpp:17-305) for every sub-tree the reference's recursion visits at one depth (FastTree.cpp:56-266).
 * lcsgpu_assign_seeds_batch: the seed sweep of FastTree::makeEvaluation (tree/FastTree.cpp:309-324) for n_jobs evaluations,
 *   each FROM SCRATCH: job g has the seeds seed_ids[seed_offsets[g] ..) and the columns col_ids[col_offsets[g] ..); per
 *   column (dist / assign, HOST, laid out like col_ids) the smallest Transform<float> distance to a seed of its job and
 *   the 0-based number of the FIRST seed attaining it -- what the reference's sweep arrives at from its first seed's row
 *   with strict '<'.  All rectangles of a call are one LCS launch per word-count class (as far as 2 GB of LCS values
 *   hold them; LCSGPU_TUNE assign_batch_kb), then one assignment launch. */
int lcsgpu_clarans_batch(lcsgpu_ctx* ctx, const int32_t* ids, const int64_t* offsets, int32_t n_jobs, int distance_kind,
                         const int32_t* n_medoids, int32_t n_fixed, float explore_fraction, int32_t num_local,
                         int32_t* medoids_out);
int lcsgpu_assign_seeds_batch(lcsgpu_ctx* ctx, const int32_t* seed_ids, const int64_t* seed_offsets, const int32_t* col_ids,
                              const int64_t* col_offsets, int32_t n_jobs, int distance_kind, float* dist, int32_t* assign);

/* Block until everything queued on the context's stream has finished. */
int lcsgpu_sync(lcsgpu_ctx* ctx);

/* Timing of the LCS kernels of the calling thread's most recent *_dev / host call on this context,
 * measured with HIP events on the stream the kernels were launched on: total milliseconds and
 * the number of kernel launches it covers.  Used by bench.py for the roofline figure. */
int lcsgpu_last_kernel_ms(lcsgpu_ctx* ctx, double* ms, int32_t* n_launches);

/* Sum of the LCS kernel milliseconds of all completed host-memory calls on this context (any thread). */
int lcsgpu_total_kernel_ms(lcsgpu_ctx* ctx, double* ms);

/* Raw stream handle (hipStream_t) of the context, for callers that enqueue their own
 * work behind the engine's. */
void* lcsgpu_stream(lcsgpu_ctx* ctx);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LCSGPU_H */
