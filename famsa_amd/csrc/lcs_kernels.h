// lcs_kernels.h -- internal interface between the C-ABI layer (lcsgpu_api / _trees / _fasttree .hip) and the
// gfx950 kernels (lcs_kernels.hip).  Not installed; the public boundary is include/lcsgpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcsgpu {

enum { MODE_RECT = 0, MODE_TRIANGLE = 1 };

// The local half of a Boruvka round FUSED into the LCS launch (triangle mode over contiguous rows / columns):
// besides (or instead of) storing LCS(row, column), a workgroup folds its 256 columns x R rows into every vertex's
// best edge to another component -- nothing but n x 8 B leaves the kernel per side.  A vertex's best edge is kept as
// ONE 64-bit record (l : 16 | length of the other endpoint : 16 | other endpoint : 32), from which MSTPrim's 128-bit
// key (distance bits, ~pack(ids); reference tree/MSTPrim.h:424-483) is a pure function -- so the global fold is a
// 64-bit compare-and-swap loop with an exact comparator (Transform<double>'s division on both records), and the
// same record is the hint for the integer pre-filter (mst_kernels.hip: l_threshold).  ~0 = none.
struct FuseArgs {
    unsigned long long* row_rec; // [n] best edge of v among u < v (its row)
    unsigned long long* col_rec; // [n] ... among u > v (its column)
    const int32_t* comp;         // [n] component labels; NULL = every vertex on its own (round 0)
    const double* pow_table;     // pow(i, 0.75) from the host's libm
    int32_t kind;                // LCSGPU_DIST_*
    int32_t on;
    // LENGTH BOUND (MSTPrim's pruning, reference tree/MSTPrim.cpp:450-467, for whole tiles): a pair of lengths (a, b) cannot
    // be closer than Transform(lcs = min(a, b)); a workgroup whose tile's smallest such bound is STRICTLY greater than the
    // distance of every record its rows and columns already hold can change none of them and skips its LCS work.
    int32_t prune;
    unsigned long long* stats;   // [0] workgroups that computed, [1] workgroups the bound let go (or NULL)
};
constexpr size_t FUSE_LDS_BYTES = 256 * 8 + 4 * 32 * 8 + 4 * 32; // per-column records + per-(wave, row) records + the bound's reductions

struct RowsArgs {
    // the uploaded sequence set (device)
    const uint8_t* tiles;      // position-major residue store, bytes = code*8
    const uint64_t* tile_base; // byte offset of each 64-sequence tile
    const uint32_t* lens;      // length per sequence
    // occurrence masks of every sequence, built once at upload: row (mask_base[i] + w) = the 32 x u64 masks
    // M[code][w] of sequence i's 64-residue word w (codes >= 20 never match: zero); words beyond a sequence: none
    const uint64_t* masks;
    const uint64_t* mask_base; // [n + 1] first row of each sequence
    // refs (bit-mask side): ids ref_ids[k] or ref_begin + k, k < n_refs
    const int32_t* ref_ids;
    const int64_t* ref_rows; // output row of ref k (else row0 + k)
    int32_t ref_begin;
    int32_t n_refs;
    // partners (streamed side): ids col_ids[c] or col_begin + c, c < n_cols
    const int32_t* col_ids;
    int32_t col_begin;
    int32_t n_cols;
    // output
    void* out;
    int64_t ld;         // RECT: out[row*ld + c]
    int64_t row0;       // first output row for contiguous refs
    int64_t out_offset; // TRIANGLE: out[row*(row-1)/2 + c - out_offset] for c < row (row as in RECT)
    int32_t elem_size;  // 2 or 4
    int32_t mode;
    int32_t refs_per_block;
    // TRIANGLE with contiguous refs: compact 1-D grid over the workgroups that have work, rows in
    // DESCENDING order (fullest first): tri_prefix[k] = first block of the k-th row from the bottom,
    // tri_prefix[tri_rows] = grid size.  NULL = plain 2-D grid (x = column block, y = ref tile).
    const int32_t* tri_prefix;
    int32_t tri_rows;
    // ... walked by DIAGONALS instead (fused launches with the length bound on: the tiles next to the diagonal pair
    // sequences of like lengths -- they make the records mature that let the far tiles go): diag_prefix[d] = first block of
    // the tiles d column blocks away from their row's last one, diag_prefix[diag_count] = grid size.  NULL = row by row.
    const int32_t* diag_prefix;
    int32_t diag_count;
    // (MODE_RECT with jobs -- lcsgpu_assign_seeds_batch: several rectangles in one launch; ref k writes the row
    // out + ref_out0[k], column c of the concatenated list at its offset c - ref_col0[k].)
    // Several triangles in one launch (lcsgpu_lcs_triangles_batch): 1-D grid, workgroup b does
    // jobs[b] = {first ref k0, ref count, first column, column limit}; rows and columns are positions
    // in the concatenated id list col_ids, ref k belongs to the list that starts at position
    // ref_col0[k] and writes its packed triangle at out + ref_out0[k].  NULL = the modes above.
    const int4* jobs;
    const int32_t* ref_col0;
    const int64_t* ref_out0;
    // Threads per workgroup: 0 / 256 = four waves sharing the refs' masks; 64 = ONE wave per workgroup (jobs only, plain refs):
    // the lists of a leaf batch are mostly shorter than 256 members, and a 256-lane workgroup over a 60-member list computes
    // with a quarter of its lanes (3 x 10^6 sequences: the leaves' 8.5 x 10^8 pairs took 0.4 s of GPU time, four times
    // their share at the large-launch rate).  Column blocks are block_threads wide then.
    int32_t block_threads;
    FuseArgs fuse; // fuse.on: MODE_TRIANGLE, contiguous columns, row == ref id; `out` may then be NULL (nothing stored)
};

const char* recolor_state();       // "on" | "off" | "failed": see csrc/Makefile, RECOLOR (the plain LCS kernels)
const char* recolor_state_fused(); // ... the translation unit of the fused instantiations
const char* kernel_id();           // sha256 prefix of the assembled listing of the plain / the fused unit
const char* kernel_id_fused();
// instantiated half-word (32-bit) counts: exact 1..32, even 34..64; 0 = the long-sequence path
int h_class(uint32_t len);
int quirk_h_class(uint32_t len);
int refs_per_block(int h, bool quirk, bool fused = false);
// ... reduced for launches that would otherwise have too few workgroups to fill the chip
int refs_per_block_for(int h, bool quirk, long n_refs, long col_blocks, bool fused = false);
hipError_t launch_rows(int h, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream);
// refs longer than 2048 residues: needs grid_x*grid_y*n_chunks_max*512 bytes of carry scratch
size_t long_carry_bytes(int grid_x, int grid_y, int n_chunks_max);
hipError_t launch_long(bool quirk, const RowsArgs& a, int grid_x, int grid_y, void* carry, int n_chunks_max,
                       hipStream_t stream);


struct RowMin {
    double dist;
    int64_t index;
};

// per-row minima over a triangle slice (see lcsgpu_row_minima_dev in include/lcsgpu.h)
// (mst_kernels.hip; minlen1024 = shortest sequence per aligned block of 1024 vertices, built at upload)
hipError_t launch_row_minima(const void* tri, int elem_size, int32_t row_begin, int32_t row_end,
                             const uint32_t* lens, const uint32_t* minlen1024, const double* pow_table, int kind,
                             RowMin* out, hipStream_t stream);


// ---- device-side Prim (tree_kernels.hip) ----
struct PrimPartial {
    double d;
    uint64_t id;
    int32_t v; // -1 = no unprocessed vertex in this workgroup
    int32_t pad;
};
struct MstEdge { // = lcsgpu_mst_edge
    int32_t from, to;
    double dist;
};
struct PrimArgs {
    const void* tri;          // lower triangle, ref = larger id
    const uint32_t* lens;
    const double* pow_table;
    const int32_t* qindex;    // per sequence: index among the orientation-sensitive ones or -1 (null if none)
    const uint32_t* q_rows;   // [n_q][n]  LCS(ref = q, partner = v)
    const uint32_t* q_cols;   // [n][n_q]  LCS(ref = v, partner = q)
    int32_t n_q;
    int32_t n;
    int32_t kind;
    int32_t n_blocks;
    double* key_d;
    uint64_t* key_id;
    uint8_t* processed;
    PrimPartial* partials;    // [2][n_blocks]
    MstEdge* edges;           // [n-1]
};
hipError_t launch_prim(const PrimArgs& a, int elem_size, hipStream_t stream);

// ---- MST by Boruvka rounds over row blocks of the triangle (mst_kernels.hip) ----
// Edge key of MSTPrim's strict total order (reference tree/MSTPrim.h:424-483): distance bits (distances are
// >= 0, so the bit patterns order like the values), then ~pack(min id, max id).  16 bytes = lcsgpu_mst_key,
// the record of the per-round exchange between the GPUs of a node.
struct MstKey {
    unsigned long long d, id;
};
struct BoruvkaArgs {
    const void* tri;            // rows [r0, r1) of the lower triangle of LCS lengths (ref = larger id):
    int64_t off;                //   element (u, v), v < u, at tri[u(u-1)/2 + v - off], off = r0(r0-1)/2
    int32_t r0, r1;
    const uint32_t* lens;
    const double* pow_table;    // pow(i, 0.75), i < pow_n, from the host's libm (the reference's values)
    int32_t pow_n, pow_in_lds;  // staged in LDS by the passes when it fits
    int32_t* comp;              // [n] component (= id of its root vertex) of every vertex -- replicated on every GPU
    int32_t* comp_next;         // [n] ... after this round
    int32_t* parent;            // [n] hooking forest over the component roots
    MstKey* row_best;           // [n] row-pass result (rows of this block only)
    uint2* row_aux;             // [n] (LCS, length of the other endpoint) of row_best: seeds the column pass's filter
    const uint32_t* minlen16;   // [ceil(n/16)]   shortest sequence of every aligned block of 16 vertices
    const uint32_t* minlen1024; // [ceil(n/1024)] ... of 1024 vertices (the passes' integer pre-filter; built at upload)
    MstKey* part;               // [n_chunks][n] column-pass partials
    MstKey* best;               // [n] this block's best edge per vertex = what a GPU contributes to the exchange
    MstKey* vbest;              // [n] best edge per vertex over all blocks (after the exchange)
    unsigned long long* cb_d;   // [n] the same per component (indexed by root)
    unsigned long long* cb_id;
    MstEdge* edges;             // [n-1] in the order the rounds find them
    int32_t* counters;          // [0] edges recorded  [1] inconsistent keys seen by the global half
    int32_t n, kind, n_chunks, rows_per_chunk;
    unsigned long long* fuse_row; // [n] the records of a local half done by the LCS launch itself (FuseArgs)
    unsigned long long* fuse_col; // [n]
    int32_t keep;                 // row_best / row_aux / part hold the LAST round's results of the same block and chunks: a record
                                  // whose edge still leaves its vertex's component stands (mst_kernels.hip, record_stands)
    int32_t crossmul;             // the exact path proves most candidates worse with one multiplication before it divides
};
hipError_t launch_boruvka_init(const BoruvkaArgs& a, hipStream_t stream);
// local half by the LCS launch (run_rows with FuseArgs over the block's rows): reset the records before it,
// turn them into a.best afterwards
hipError_t launch_boruvka_fuse_reset(const BoruvkaArgs& a, bool keep, hipStream_t stream);
hipError_t launch_boruvka_fuse_fold(const BoruvkaArgs& a, hipStream_t stream);
// local half of a round: a.best[v] = best edge of v to another component among the pairs of this row block
hipError_t launch_boruvka_best(const BoruvkaArgs& a, int elem_size, hipStream_t stream);
// global half: `gathered` = n_parts x n keys (every block's a.best); per-component minima, hooking, relabel.
// The caller swaps comp / comp_next afterwards and reads counters[0].
hipError_t launch_boruvka_merge(const BoruvkaArgs& a, const MstKey* gathered, int n_parts, hipStream_t stream);

// ---- device-side UPGMA (tree_kernels.hip) ----
struct UpgmaArgs {
    float* D;             // float distances (updated in place): the packed lower triangle, or -- square -- the full
                          // symmetric n x n matrix, in which both rows a merge reads are contiguous
    int32_t square;
    int64_t ld;           // square: floats from one row of D to the next (n; 2n in the slot layout of upgma_batch_kernels.hip)
    float* min_dist;      // [n]
    uint32_t* nearest;    // [n]
    uint32_t* node_index; // [n]
    float* part_d;        // [2][n_blocks] per-workgroup minima of the new row, alternating between merges
    uint32_t* part_j;
    float* bm_d;          // [n_blocks] first minimum of min_dist over each workgroup's 256 rows,
    uint32_t* bm_j;       //            its row
    uint32_t* bm_near;    //            and that row's nearest
    uint32_t* sel;        // 64 words: (Lmin, Rmin) of the merges, error flag, the touched workgroups' other minima
                          // (tree_kernels.hip, UPGMA_SEL_EXCL)
    int32_t* left;        // [n-1] children of the internal nodes
    int32_t* right;
    int32_t n;
    int32_t n_blocks;
};
// float distances of the rows [r0, r1) from their LCS values (lcs = the packed triangle from row r0 on; square layout: r0
// a multiple of 32), block after block; then the initial row minima; then the n launches of the merge steps
hipError_t launch_upgma_distances(const UpgmaArgs& a, const void* lcs, int elem_size, const uint32_t* lens, const float* pow_f32,
                                  int kind, int r0, int r1, hipStream_t stream);
hipError_t launch_upgma_init(const UpgmaArgs& a, hipStream_t stream);
hipError_t launch_upgma_steps(const UpgmaArgs& a, bool modified, hipStream_t stream);

// ---- several merges per launch (upgma_batch_kernels.hip; symmetric-matrix layout only) ----
constexpr int UPGMA_BATCH_MAX = 32;  // merges per batch at most (template instances: 8, 16, 32)
constexpr int UPGMA_BATCH_CAND = 64; // entries of the sorted order handed to the next batch's walk (2 x the largest batch)
struct UpgmaBatchArgs {
    float* D;             // D[row * ld + slot]: n rows x ld slots (ld >= 2n - 1; see upgma_batch_kernels.hip, LAYOUT)
    int64_t ld;
    uint32_t* slot_of;    // [n]  the slot a row's cluster sits in
    uint32_t* row_of;     // [ld] the row a slot stands for, NONE when the slot is dead / not used yet
    float* min_dist;      // [n]   as UpgmaArgs
    uint32_t* nearest;    // [n]
    uint32_t* node_index; // [n]
    int32_t* left;        // [n-1]
    int32_t* right;
    int32_t n;
    int32_t n_blocks;     // ceil(ld / 256): workgroups that cover every slot
    uint2* sorted0;       // [n + 1] the active rows as (min_dist bits, row), ascending; batch parity 0 reads it, 1 writes it
    uint2* sorted1;       //         ... and the other way round
    uint32_t* pos;        // [n] where a row's entry sits in the current order
    uint4* cand;          // [UPGMA_BATCH_CAND] the first entries of the current order with their rows' nearest
    uint32_t* state;      // [2][8]: merges committed, entries of the order, error, batches cut short, the next free slot -- by batch parity
    uint32_t* remap;      // [ld] compaction: where a live slot moves to (upgma_compact_*_kernel)
    uint32_t* hdr;        // [512] the pending batch: count, then per merge (L, R, key bits, creator of R, positions, slots)
    uint32_t* rec;        // [8 + 4 K + K^2] what the resolve kernel found: V, per merge (new min_dist, nearest, die), cross entries
    float* side;          // [UPGMA_BATCH_MAX][ld] the rows the pending batch creates, along the slots
    float* part_d;        // [UPGMA_BATCH_MAX][n_blocks] per-workgroup first minima of those rows
    uint32_t* part_j;
};
hipError_t launch_upgma_batch_init(const UpgmaBatchArgs& a, hipStream_t stream);
// `count` batches, the first of them batch number `first`; slots_used = an upper bound of the next free slot before them
hipError_t launch_upgma_batches(const UpgmaBatchArgs& a, bool modified, int k, int first, int count, long long slots_used, hipStream_t stream);
// Compaction between two batches (the next one has parity `parity`): the live slots move to the front in order, the
// matrix rows are packed in place; slots_used as above.  The next free slot afterwards = the number of live clusters.
hipError_t launch_upgma_compact(const UpgmaBatchArgs& a, int parity, long long slots_used, hipStream_t stream);

// ---- device-side neighbour joining (tree_kernels.hip) ----
struct NjArgs {
    float* D;         // float distance triangle (updated in place)
    float* sum;       // [n] sum of distances per cluster row
    float* tmp;       // [n] new distances of the current merge
    float* part_q;    // [n] per-row minima of q
    int32_t* part_i;  // [n]
    int32_t* node;    // [n] tree node id held by the row
    uint8_t* active;  // [n]
    int32_t* sel;     // [0] mi, [1] mj, [2] degenerate flag
    int32_t* left;    // [n-1]
    int32_t* right;
    int32_t n;
};
hipError_t launch_nj(const NjArgs& a, hipStream_t stream);
hipError_t launch_nj_init(const NjArgs& a, hipStream_t stream);
hipError_t launch_float_distances(const void* lcs, int elem_size, const uint32_t* lens, const float* pow_f32, int kind,
                                  int n, float* D, hipStream_t stream);

// seed assignment over an LCS rectangle in HBM (tree_kernels.hip); dist / assign are read and updated
hipError_t launch_assign_seeds(const void* lcs, int elem_size, int64_t ld, const int32_t* seed_ids, int32_t n_seeds,
                               const int32_t* col_ids, int32_t n_cols, const uint32_t* lens, const float* pow_f32,
                               int kind, int first_k, float* dist, int32_t* assign, hipStream_t stream);

// ---- the uploaded set's device form (upload_kernels.hip) ----
// tiles / quirk flags from the packed codes; flags[0] |= 1 if a symbol code >= 32 was met
hipError_t launch_build_set(const uint8_t* codes, const uint64_t* offsets, const int32_t* order, const uint64_t* tile_base, int32_t n,
                            uint8_t* tiles, uint8_t* quirk, int32_t* flags, const uint64_t* mask_base, uint64_t* masks,
                            hipStream_t stream);

// ---- device-side CLARANS (clarans_kernels.hip) ----
constexpr int CLARANS_MAX_MEDOIDS = 1024;
constexpr int CLARANS_MAX_NONMEDOIDS = 2048; // every position's state in the registers of one workgroup
struct ClaransArgs {
    const float* D;      // the sample members' float distances, full symmetric matrix D[i * n_elems + j]
    float* DMt;          // [n_medoids][n_elems] distance of the member at a position to the medoid in a slot
    int32_t* cand;       // [n_elems] permutation of the members; positions < n_medoids are the medoids
    float4* st;          // [n_elems] by position: {d(nearest), d(second), slot(nearest), slot(second)}
    const int32_t* draws; // pre-drawn positions xx of the steps (the position generator's output)
    float* cost_log;     // [1 + n_elems] addends of the running cost, in the reference's order
    int32_t* state;      // the search's state block (clarans_kernels.hip, ST_*): next draw, done, accepts, cost, window offset, statistics
    int32_t n_elems, n_medoids, n_fixed, draws_len;
    int32_t corrected;   // steps without an accept that end a local search (Clustering.cpp:21-29)
};

// ---- -dist_export rows as text, on the device (text_kernels.hip) ----
struct TextArgs {
    const void* lcs;         // the block's LCS rectangle: row r (sequence row_begin + r) x ld, elem_size 2 | 4
    int64_t ld;
    const int32_t* where;    // [n] column of sequence j inside the rectangle (the rectangle's columns are in length order); NULL = j
    const uint32_t* lens;
    const double* pow_f64;   // pow(i, 0.75) from the host's libm
    const char* ids;         // the sequences' names without '>' ...
    const uint64_t* id_off;  // ... name i = ids[id_off[i] .. id_off[i + 1])
    uint32_t* seg_len;       // [n_rows x segs] bytes of a segment, then its offset inside the row
    uint32_t* row_len;       // [n_rows]
    unsigned long long* row_start; // [n_rows + 1] offset of a row inside the block's text; [n_rows] = the block's bytes
    char* out;               // the block's text
    int32_t elem_size;
    int32_t kind;            // LCSGPU_DIST_* through Transform<double> -> float, or 2: Transform<float, pairwise_identity>
    int32_t row_begin, n_rows;
    int32_t n;
    int32_t square;          // row i holds n values, else the i values j < i
    int32_t segs;            // segments (text_segment_values() values each) per row
};
hipError_t launch_text_block(const TextArgs& t, hipStream_t stream);
int text_segment_values();
int text_max_value_bytes();

} // namespace lcsgpu
