// mst_kernels.hip -- the minimum spanning tree of the uploaded set by Boruvka rounds over ROW BLOCKS of
// the LCS triangle in HBM: one GPU holding the whole triangle, or the GPUs of a node holding one row
// block each (the N x N pair space tiled by row block).
//
// MSTPrim (reference tree/MSTPrim.cpp:356-533) orders edges by the strict total order
// (d, ~pack(min id, max id)) -- smaller distance first, then the larger packed id (cpp:493-509) -- so the
// MST is unique and any exact MST algorithm yields the reference's edge set; Prim's insertion order from
// vertex 0 (cpp:372-391) is then a walk over those n-1 edges, done by the caller on the host, and the
// dendrogram (cpp:784-833) follows from that order.  Prim itself needs n-1 dependent steps; Boruvka needs
// <= log2(n) rounds of streaming passes (2 B per pair).  A round has a LOCAL half and a GLOBAL half:
//   local  (per row block [r0, r1), no communication): every vertex's best edge to another component
//          among the pairs the block holds -- the row part (u < v, v in the block: v's own contiguous row)
//          and the column part (u > v, u in the block: lanes = consecutive columns, walking down the
//          rows, coalesced) -> best[v], n x 16 B.  Round 0 of this is the "per-row minima" of the
//          north star, completed by the column part.
//   global (replicated on every GPU after the exchange of the best[] arrays -- n x 16 B per GPU, an
//          all-gather): fold the blocks' keys, every component's best edge (two 64-bit atomic-min phases:
//          distance bits, then id), hook each component to the other end of its edge (mutual choices:
//          the smaller root stays a root and the edge is recorded once), relabel the vertices.
// The component labels are a pure function of the exchanged keys, so every GPU derives the same ones.
//
// Distances: Transform<double, kind> (reference tree/AbstractTreeGenerator.hpp:28-82) exactly -- host-built
// pow table + IEEE f64 division -- but only for candidates that can win: one integer comparison of the LCS
// length against a threshold derived from the current best edge (l_threshold below) proves most candidates
// worse than the best, and component labels, lengths, the table look-up, the f64 division and the 128-bit
// compare are skipped for them.  History of the two passes at n = 100 000 (10 GB of u16 each) is in DESIGN.md.
//
// Valid when d(u, v) does not depend on which endpoint is the ref: always for the triangle's own
// orientation (what SLINK sees), and for MSTPrim's orientation when no uploaded sequence is orientation
// sensitive (SURVEY note Q); otherwise the caller keeps the Prim kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "lcs_kernels.h"

namespace lcsgpu {

namespace {

constexpr unsigned long long NO_D = 0x7fefffffffffffffull; // bits of DBL_MAX: "no candidate"
constexpr unsigned long long NO_ID = ~0ull;

__device__ __forceinline__ bool key_less(unsigned long long d1, unsigned long long i1, unsigned long long d2,
                                         unsigned long long i2)
{
    return d1 < d2 || (d1 == d2 && i1 < i2); // distances are >= 0: their bit patterns order like the values
}

__device__ __forceinline__ unsigned long long pack_ids(uint32_t a, uint32_t b) // ids_to_uint64, tree/MSTPrim.h:432-439
{
    return a < b ? ((unsigned long long)a << 32) + b : ((unsigned long long)b << 32) + a;
}

// a lane's running best: exact key, the (l, indel) it came from (ties on both are settled on the ids alone) and the
// length of the edge's other endpoint (what the integer pre-filter needs besides l)
struct Best {
    unsigned long long d = NO_D, id = NO_ID;
    uint32_t l = ~0u, indel = ~0u, len_o = 0;
};

// pow(i, 0.75) for the exact path: from LDS when the table fits -- the exact path then touches no global
// memory for it -- else from HBM
template <bool IN_LDS>
struct PowTable {
    const double* p;
    __device__ __forceinline__ double operator()(uint32_t i) const { return p[i]; }
};
template <bool IN_LDS>
__device__ __forceinline__ PowTable<IN_LDS> stage_pow_table(const BoruvkaArgs& a, double* smem)
{
    if (!IN_LDS) return PowTable<IN_LDS>{a.pow_table};
    for (int i = threadIdx.x; i < a.pow_n; i += 256) smem[i] = a.pow_table[i];
    __syncthreads();
    return PowTable<IN_LDS>{smem};
}

// The pre-filter: ONE integer comparison of the LCS length.  For a fixed vertex v, d(l, S) = f(S - 2l) / l with
// S = len_v + len_other is decreasing in l and increasing in S, and along a level line d = const the slope
// dl/dS = 0.75 l / (1.5 l + indel)  (indel^0.75 / l)  resp.  l / S  (indel / l)  never exceeds 1/2.  So with a best
// edge (l_b, len_b) in hand, a candidate whose other endpoint is at least `minlen` long can reach the best's
// distance only with  l >= l_b - max(0, len_b - minlen) / 2;  every l at least one below that bound is worse by
// a factor >= 1 + 1/65535, far beyond the roundings of the table and the division, so dropping
//     l < l_b - ceil(max(0, len_b - minlen) / 2)
// drops no candidate that could win or tie.  minlen = the shortest sequence of the batch (minlen16 / minlen1024,
// built once per tree), so the bulk of a pass costs a max-reduction and a compare per batch; component labels,
// lengths, the table look-up, the f64 division and the 128-bit key compare are touched by the survivors only.
__device__ __forceinline__ uint32_t l_threshold(uint32_t l_b, uint32_t len_b, uint32_t minlen)
{
    if (l_b == ~0u) return 0; // no best yet: everything goes the exact way
    const uint32_t slack = len_b > minlen ? (len_b - minlen + 1) >> 1 : 0;
    return l_b > slack ? l_b - slack : 0;
}

// The exact comparison: Transform<double, KIND> (hpp:28-82) and MSTPrim's key order.  A candidate with the
// best's own (l, indel) has the best's distance bit for bit: only the ids decide, no division.  Returns true
// if the best's (l, length) changed.
//
// Before the division, one multiplication (CROSSMUL): with p = the numerator (a table entry or an integer: 0 or >= 1, so no
// quotient is subnormal),
// the quotient q = p / l rounds to fl(q) >= q (1 - 2^-53).  If p > fl(fl(d_b * l) * (1 + 2^-50)) then, the two roundings of
// the right-hand side being at most 2^-52 relative together, p > d_b l (1 + 2^-51), so q > d_b (1 + 2^-51) and
// fl(q) > d_b (1 + 2^-51)(1 - 2^-53) > d_b: the candidate is strictly worse and never reaches the ids.  (No best yet:
// d_b = DBL_MAX, the product is inf or DBL_MAX-sized and nothing is dropped.)  On ragged sets the integer pre-filter lets
// most candidates through (its slack is half the difference of lengths); this test ends them without the ~30
// instructions of an IEEE f64 division.
__device__ __forceinline__ bool exact_update_p(Best& b, uint32_t l, uint32_t indel, double p, uint32_t lo, uint32_t hi,
                                               uint32_t len_other, bool crossmul)
{
    const unsigned long long id = ~(((unsigned long long)lo << 32) + hi);
    if (l == b.l && indel == b.indel) {
        if (id < b.id) b.id = id;
        return false;
    }
    double d;
    if (l == 0) d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0), hpp:61,73
    else {
        if (crossmul && p > (__longlong_as_double((long long)b.d) * (double)l) * (1.0 + 0x1p-50)) return false;
        d = p / (double)l;
    }
    const unsigned long long db = (unsigned long long)__double_as_longlong(d);
    if (!key_less(db, id, b.d, b.id)) return false;
    b.d = db;
    b.id = id;
    b.l = l;
    b.indel = indel;
    b.len_o = len_other;
    return true;
}

template <int KIND, typename PW>
__device__ __forceinline__ bool exact_update(const PW& pw, Best& b, uint32_t l, uint32_t indel, uint32_t lo, uint32_t hi,
                                             uint32_t len_other, bool crossmul = true)
{
    return exact_update_p(b, l, indel, KIND == 1 ? pw(indel) : (double)indel, lo, hi, len_other, crossmul);
}

// A record of the LAST round stands in this one when its edge still leaves its vertex's component: components only
// merge, so the edges that cross now are a subset of those that crossed then, and the smallest of then that is still
// among them is the smallest of now.  (Same subset of pairs both times: the same row, the same chunk of the same block.)
__device__ __forceinline__ bool record_stands(const BoruvkaArgs& a, const MstKey& k, int v, int cv)
{
    if (k.id == NO_ID) return false;
    const unsigned long long packed = ~k.id;
    const uint32_t lo = (uint32_t)(packed >> 32), hi = (uint32_t)packed;
    const uint32_t other = lo == (uint32_t)v ? hi : lo;
    return other < (uint32_t)a.n && a.comp[other] != cv;
}

// smallest key of the wave, in every lane
__device__ __forceinline__ void wave_min_key(unsigned long long& d, unsigned long long& id)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long d2 = __shfl_xor(d, o, 64), i2 = __shfl_xor(id, o, 64);
        if (key_less(d2, i2, d, id)) { d = d2; id = i2; }
    }
}

} // namespace

__global__ __launch_bounds__(256) void boruvka_init_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) a.comp[v] = v;
    if (v == 0) {
        a.counters[0] = 0; // edges recorded so far
        a.counters[1] = 0; // set by the global half when the gathered keys are inconsistent
    }
}

// The two passes stream the block's triangle once each (2 B per pair) and are meant to be bound by that stream:
// per batch of 16 elements a lane does 15 max + 1 compare against the integer threshold above; only batches with a
// survivor look at component labels and lengths (gathers) and at the exact key.  (History: the float pre-filter of
// the first version cost ~14 VALU per element plus per-element scalar work -- the CU's scalar unit was 80% busy --
// and read comp[] / lens[] for every column: row pass 2.85 ms, column pass 3.8 ms per 10 GB at n = 100 000.)
constexpr int COLS_PER_WG = 256; // columns of one workgroup of the column pass
constexpr int ROWS_PER_WG = 4;   // rows of one workgroup of the row pass: one per wave
typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// best edge of the vertex v (one WAVE per row, the block's longest rows first) among u < v (row v of the triangle)
// to another component; lanes stride over the columns, 16 x 64 columns per batch, two batches of loads in flight.
// All lanes work for the same vertex, so the filter's threshold is the wave's: after every update the best of the
// wave is found and its (l, length) broadcast.
template <typename T, int KIND, bool POW_LDS>
__global__ __launch_bounds__(256) void boruvka_row_kernel(BoruvkaArgs a)
{
    extern __shared__ double s_pow[];
    constexpr int UNR = 16, STEP = 64 * UNR;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, and known to be: the row, its base and the loop bounds stay in SGPRs
    const PowTable<POW_LDS> pw = stage_pow_table<POW_LDS>(a, s_pow);
    const int v = a.r1 - 1 - (blockIdx.x * ROWS_PER_WG + wave);
    if (v < a.r0) return;
    if (v < 1) { // row 0: nothing below it
        if (lane == 0) {
            a.row_best[v] = MstKey{NO_D, NO_ID};
            a.row_aux[v] = make_uint2(~0u, 0u);
        }
        return;
    }
    const int cv = a.comp[v];
    if (a.keep && record_stands(a, a.row_best[v], v, cv)) return; // (uniform: every lane looks at the same record)
    const bool crossmul = a.crossmul != 0;
    const uint32_t len_v = a.lens[v];
    const T* row = (const T*)a.tri + ((int64_t)v * (v - 1) / 2 - a.off);
    const int last = v - 1;
    Best b;
    uint32_t wl = ~0u, wlen = 0; // (l, length) of the wave's best edge so far
    T l[2][UNR];
    auto request = [&](int ub, int s) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) l[s][k] = row[min(ub + lane + 64 * k, last)];
    };
    auto evaluate = [&](int ub, int s) {
        const uint32_t thr = l_threshold(wl, wlen, a.minlen1024[ub >> 10]);
        uint32_t m = l[s][0];
#pragma unroll
        for (int k = 1; k < UNR; ++k) m = max(m, (uint32_t)l[s][k]);
        if (__builtin_amdgcn_ballot_w64(m >= thr) == 0) return;
        bool changed = false;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int u = ub + lane + 64 * k;
            const uint32_t lv = l[s][k];
            if (lv >= thr && u < v) {
                if (a.comp[u] != cv) {
                    const uint32_t lu = a.lens[u];
                    changed |= exact_update<KIND>(pw, b, lv, len_v + lu - 2u * lv, (uint32_t)u, (uint32_t)v, lu, crossmul);
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(changed) != 0) {
            unsigned long long d = b.d, id = b.id;
            wave_min_key(d, id);
            const unsigned long long who = __builtin_amdgcn_ballot_w64(b.d == d && b.id == id);
            const int src = __ffsll((long long)who) - 1;
            wl = (uint32_t)__builtin_amdgcn_readlane((int)b.l, src);
            wlen = (uint32_t)__builtin_amdgcn_readlane((int)b.len_o, src);
        }
    };
    request(0, 0);
    for (int ub = 0; ub < v; ub += 2 * STEP) {
        request(ub + STEP, 1); // past the row's end: clamped, never evaluated
        evaluate(ub, 0);
        request(ub + 2 * STEP, 0);
        if (ub + STEP < v) evaluate(ub + STEP, 1);
    }
    unsigned long long d = b.d, id = b.id;
    wave_min_key(d, id);
    const unsigned long long who = __builtin_amdgcn_ballot_w64(b.d == d && b.id == id);
    const int src = __ffsll((long long)who) - 1;
    const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b.l, src);
    const uint32_t blen = (uint32_t)__builtin_amdgcn_readlane((int)b.len_o, src);
    if (lane == 0) {
        a.row_best[v] = MstKey{d, id};
        a.row_aux[v] = make_uint2(bl, blen);
    }
}

// best edge of vertex v (lane = column) among the rows u > v of one row chunk of the block -> part[chunk][v].
// The row index is uniform across the workgroup: what the survivors need of comp[u] / lens[u] are scalar loads.
// A lane of the block's own rows starts from the row pass's result for its vertex (the row pass runs first):
// only candidates that beat it matter, and its threshold is mature from the first batch on.
template <typename T, int KIND, bool POW_LDS>
__global__ __launch_bounds__(256) void boruvka_col_kernel(BoruvkaArgs a)
{
    constexpr int UNR = 16; // rows per batch (two batches in flight)
    // Which (column block, chunk) this workgroup does.  Rows of the triangle start at any 2-byte offset, so the 512 B a
    // workgroup reads of a row share their first and last cache line with the neighbouring column blocks: fetched once if
    // the neighbours run on the same XCD at about the same time, once per L2 otherwise (round 3 measured 1.34 x the
    // triangle per pass).  Workgroups go to the XCDs round-robin in dispatch order; the dispatch index is rearranged so
    // that every XCD gets runs of XCD_RUN neighbouring column blocks (runs themselves still round-robin: the work per
    // column block grows along the grid, an XCD must not get one contiguous eighth of it).
    constexpr unsigned XCD_RUN = 8;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const unsigned gx = gridDim.x, total = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
        const unsigned whole = total - total % (8u * XCD_RUN); // the tail keeps its place
        if (lin < whole) {
            const unsigned xcd = lin & 7u, slot = lin >> 3;
            const unsigned moved = ((slot / XCD_RUN) * 8u + xcd) * XCD_RUN + slot % XCD_RUN;
            bx = (int)(moved % gx);
            by = (int)(moved / gx);
        }
    }
    const int c0 = bx * COLS_PER_WG;
    const int v = c0 + threadIdx.x;
    const int chunk = by;
    const int u0 = a.r0 + chunk * a.rows_per_chunk, u1 = min(a.r1, u0 + a.rows_per_chunk);
    if (u0 >= u1 || c0 + 1 >= u1) return; // empty chunk, or every column of this workgroup lies at or above its last row: fold skips it
    extern __shared__ double s_pow[];
    const PowTable<POW_LDS> pw = stage_pow_table<POW_LDS>(a, s_pow);
    if (v >= a.n) return;
    Best b;
    const int cv = a.comp[v];
    // the last round's partial of this chunk (the smaller of the row's and the chunk's own) stands: this lane is done
    if (a.keep && record_stands(a, a.part[(size_t)chunk * a.n + v], v, cv)) return;
    const bool crossmul = a.crossmul != 0;
    const uint32_t len_v = a.lens[v];
    const T* tri = (const T*)a.tri - a.off;
    if (v >= a.r0 && v < a.r1) {
        const MstKey k = a.row_best[v];
        if (k.id != NO_ID) {
            const uint2 aux = a.row_aux[v];
            b.d = k.d;
            b.id = k.id;
            b.l = aux.x;
            b.len_o = aux.y;
            b.indel = len_v + aux.y - 2u * aux.x;
        }
    }

    // rows [ua, ub): one by one, with the range test (the diagonal block and the chunk's last rows)
    auto plain_rows = [&](int ua, int ub) {
        for (int u = ua; u < ub; ++u) {
            const int cu = a.comp[u];
            const uint32_t len_u = a.lens[u];
            if (u > v && cu != cv) {
                const uint32_t lv = tri[(int64_t)u * (u - 1) / 2 + v];
                if (lv >= l_threshold(b.l, b.len_o, len_u))
                    exact_update<KIND>(pw, b, lv, len_u + len_v - 2u * lv, (uint32_t)v, (uint32_t)u, len_u, crossmul);
            }
        }
    };
    const int first = max(u0, c0 + 1);
    const int bulk0 = min(u1, max(first, c0 + COLS_PER_WG)); // from here on every lane's column lies below the row
    plain_rows(first, bulk0);

    T l[2][UNR];
    // rows ub .. ub + UNR - 1 (all < u1): ONE 64-bit row base per batch (scalar), then 32-bit offsets that grow by the
    // row length -- the per-row 64-bit u(u-1)/2 cost ~15 scalar instructions each, and the CU's one scalar unit,
    // shared by the waves of all four SIMDs, was what the pass waited for
    auto request = [&](int ub, int s) {
        const char* base = (const char*)(tri + (int64_t)ub * (ub - 1) / 2);
        uint32_t off = (uint32_t)sizeof(T) * (uint32_t)v;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            l[s][k] = *(const T*)(base + off);
            off += (uint32_t)sizeof(T) * (uint32_t)(ub + k);
        }
    };
    auto evaluate = [&](int ub, int s) { // a full batch of the bulk: rows ub .. ub + UNR - 1 < u1, all below every lane's column
        const uint32_t mb = min(a.minlen16[ub >> 4], a.minlen16[(ub + UNR - 1) >> 4]);
        uint32_t thr = l_threshold(b.l, b.len_o, mb);
        uint32_t m = l[s][0];
#pragma unroll
        for (int k = 1; k < UNR; ++k) m = max(m, (uint32_t)l[s][k]);
        if (m >= thr) {
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const uint32_t lv = l[s][k];
                if (lv >= thr) {
                    const int u = ub + k;
                    if (a.comp[u] != cv) {
                        const uint32_t len_u = a.lens[u];
                        if (exact_update<KIND>(pw, b, lv, len_u + len_v - 2u * lv, (uint32_t)v, (uint32_t)u, len_u, crossmul))
                            thr = l_threshold(b.l, b.len_o, mb);
                    }
                }
            }
        }
    };
    int ub = bulk0;
    const int n_pairs = (u1 - bulk0) / (2 * UNR);
    if (n_pairs > 0) {
        request(ub, 0);
        for (int it = 0; it < n_pairs; ++it, ub += 2 * UNR) {
            request(ub + UNR, 1);
            evaluate(ub, 0);
            if (it + 1 < n_pairs) request(ub + 2 * UNR, 0);
            evaluate(ub + UNR, 1);
        }
    }
    plain_rows(ub, u1);
    a.part[(size_t)chunk * a.n + v] = MstKey{b.d, b.id};
}

// ---- per-row minima over a triangle slice (lcsgpu_row_minima_dev) -----------------------------------------
// Row i's nearest neighbour among j < i with MSTPrim's edge order (reference tree/MSTPrim.cpp:493-509: smaller d,
// then the larger j) = the row pass above without component labels: one wave per row, the integer pre-filter,
// the exact f64 key for survivors only.  HBM-bound: reads the slice once (round 1 computed the f64 distance of
// every pair: 11.7 ms per 10 GB).
template <typename T, int KIND>
__global__ __launch_bounds__(256) void row_minima_kernel(const T* __restrict__ tri, int32_t row_begin, int32_t row_end,
                                                         const uint32_t* __restrict__ lens,
                                                         const uint32_t* __restrict__ minlen1024,
                                                         const double* __restrict__ pow_table, RowMin* __restrict__ out)
{
    constexpr int UNR = 16, STEP = 64 * UNR;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, and known to be: the row, its base and the loop bounds stay in SGPRs
    const PowTable<false> pw{pow_table};
    const int v = row_end - 1 - (blockIdx.x * ROWS_PER_WG + wave); // the slice's longest rows first
    if (v < row_begin) return;
    RowMin* o = out + (v - row_begin);
    if (v < 1) {
        if (lane == 0) {
            o->dist = 1.7976931348623157e308; // DBL_MAX: an empty row
            o->index = -1;
        }
        return;
    }
    const uint32_t len_v = lens[v];
    const int64_t off = (int64_t)row_begin * (row_begin - 1) / 2;
    const T* row = tri + ((int64_t)v * (v - 1) / 2 - off);
    const int last = v - 1;
    Best b;
    uint32_t wl = ~0u, wlen = 0;
    T l[2][UNR];
    auto request = [&](int ub, int s) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) l[s][k] = row[min(ub + lane + 64 * k, last)];
    };
    auto evaluate = [&](int ub, int s) {
        const uint32_t thr = l_threshold(wl, wlen, minlen1024[ub >> 10]);
        uint32_t m = l[s][0];
#pragma unroll
        for (int k = 1; k < UNR; ++k) m = max(m, (uint32_t)l[s][k]);
        if (__builtin_amdgcn_ballot_w64(m >= thr) == 0) return;
        bool changed = false;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int u = ub + lane + 64 * k;
            const uint32_t lv = l[s][k];
            if (lv >= thr && u < v) {
                const uint32_t lu = lens[u];
                changed |= exact_update<KIND>(pw, b, lv, len_v + lu - 2u * lv, (uint32_t)u, (uint32_t)v, lu);
            }
        }
        if (__builtin_amdgcn_ballot_w64(changed) != 0) {
            unsigned long long d = b.d, id = b.id;
            wave_min_key(d, id);
            const unsigned long long who = __builtin_amdgcn_ballot_w64(b.d == d && b.id == id);
            const int src = __ffsll((long long)who) - 1;
            wl = (uint32_t)__builtin_amdgcn_readlane((int)b.l, src);
            wlen = (uint32_t)__builtin_amdgcn_readlane((int)b.len_o, src);
        }
    };
    request(0, 0);
    for (int ub = 0; ub < v; ub += 2 * STEP) {
        request(ub + STEP, 1);
        evaluate(ub, 0);
        request(ub + 2 * STEP, 0);
        if (ub + STEP < v) evaluate(ub + STEP, 1);
    }
    unsigned long long d = b.d, id = b.id;
    wave_min_key(d, id);
    if (lane == 0) {
        o->dist = __longlong_as_double((long long)d);
        o->index = (int64_t)((~id) >> 32); // id = ~pack(j, i), j < i
    }
}

hipError_t launch_row_minima(const void* tri, int elem_size, int32_t row_begin, int32_t row_end, const uint32_t* lens,
                             const uint32_t* minlen1024, const double* pow_table, int kind, RowMin* out,
                             hipStream_t stream)
{
    const int rows = row_end - row_begin;
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), block(256);
#define ROW_MIN(T, K)                                                                                        \
    hipLaunchKernelGGL((row_minima_kernel<T, K>), grid, block, 0, stream, (const T*)tri, row_begin, row_end, \
                       lens, minlen1024, pow_table, out)
    if (elem_size == 2) {
        if (kind == 1) ROW_MIN(uint16_t, 1); else ROW_MIN(uint16_t, 0);
    } else {
        if (kind == 1) ROW_MIN(uint32_t, 1); else ROW_MIN(uint32_t, 0);
    }
#undef ROW_MIN
    return hipGetLastError();
}

// this block's best edge per vertex: its row part (rows of the block) and the column partials that exist
__global__ __launch_bounds__(256) void boruvka_fold_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    unsigned long long bd = NO_D, bi = NO_ID;
    if (v >= a.r0 && v < a.r1) {
        const MstKey k = a.row_best[v];
        bd = k.d;
        bi = k.id;
    }
    const int c0 = v / COLS_PER_WG * COLS_PER_WG;
    for (int c = 0; c < a.n_chunks; ++c) {
        const int u0 = a.r0 + c * a.rows_per_chunk, u1 = min(a.r1, u0 + a.rows_per_chunk);
        if (u0 >= u1 || c0 + 1 >= u1) continue; // not written (see boruvka_col_kernel)
        const MstKey k = a.part[(size_t)c * a.n + v];
        if (key_less(k.d, k.id, bd, bi)) { bd = k.d; bi = k.id; }
    }
    a.best[v] = MstKey{bd, bi};
}

// ---- the local half done by the LCS launch itself (lcs_kernels.hip, FuseArgs): reset / conversion of the records ----
// keep: the records of the last round stay where their edge still leaves the vertex's component (a.comp = this round's labels)
__global__ __launch_bounds__(256) void boruvka_fuse_reset_kernel(BoruvkaArgs a, int keep)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    unsigned long long r = ~0ull, c = ~0ull;
    if (keep) {
        const int cv = a.comp[v];
        r = a.fuse_row[v];
        c = a.fuse_col[v];
        if (r != ~0ull && a.comp[(uint32_t)r] == cv) r = ~0ull;
        if (c != ~0ull && a.comp[(uint32_t)c] == cv) c = ~0ull;
    }
    a.fuse_row[v] = r;
    a.fuse_col[v] = c;
}

// (l, length of the other endpoint, other endpoint) -> MSTPrim's key, with the arithmetic of exact_update above
__device__ __forceinline__ MstKey fuse_key(const BoruvkaArgs& a, unsigned long long rec, int v)
{
    if (rec == ~0ull) return MstKey{NO_D, NO_ID};
    const uint32_t l = (uint32_t)(rec >> 48), u = (uint32_t)rec;
    const uint32_t indel = a.lens[v] + a.lens[u] - 2u * l;
    double d;
    if (l == 0) d = 1.7976931348623155e308;
    else if (a.kind == 1) d = a.pow_table[indel] / (double)l;
    else d = (double)indel / (double)l;
    const uint32_t lo = u < (uint32_t)v ? u : (uint32_t)v, hi = u < (uint32_t)v ? (uint32_t)v : u;
    return MstKey{(unsigned long long)__double_as_longlong(d), ~(((unsigned long long)lo << 32) + hi)};
}

__global__ __launch_bounds__(256) void boruvka_fuse_fold_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    const MstKey r = fuse_key(a, a.fuse_row[v], v), c = fuse_key(a, a.fuse_col[v], v);
    a.best[v] = key_less(c.d, c.id, r.d, r.id) ? c : r;
}

__global__ __launch_bounds__(256) void boruvka_reset_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    a.cb_d[v] = NO_D;
    a.cb_id[v] = NO_ID;
    a.parent[v] = v;
}

// Per-component minima with few global atomics.  In the late rounds thousands of vertices share a component, and one
// atomicMin per vertex on a handful of addresses serialises at the L2 (193 us per round on average at n = 100 000).  A
// workgroup first combines its 256 vertices in LDS -- a 512-entry open-addressing table keyed by the component label,
// LDS atomics -- and only the table's occupied entries go to global memory: at most one atomic per (workgroup, component).
constexpr int CB_SLOTS = 512;
__device__ __forceinline__ void cb_table_min(int* s_key, unsigned long long* s_val, int c, unsigned long long value)
{
    unsigned h = ((unsigned)c * 2654435761u) >> 23; // 9 bits
    for (;;) {
        const int seen = atomicCAS(&s_key[h], -1, c);
        if (seen == -1 || seen == c) {
            atomicMin(&s_val[h], value);
            return;
        }
        h = (h + 1) & (CB_SLOTS - 1); // (256 keys at most in 512 slots: a free slot is always found)
    }
}

// fold the blocks' keys into vbest[v]; first atomic phase of the per-component minimum
__global__ __launch_bounds__(256) void boruvka_gather_kernel(BoruvkaArgs a, const MstKey* gathered, int n_parts)
{
    __shared__ int s_key[CB_SLOTS];
    __shared__ unsigned long long s_val[CB_SLOTS];
    for (int i = threadIdx.x; i < CB_SLOTS; i += 256) {
        s_key[i] = -1;
        s_val[i] = NO_D;
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) {
        unsigned long long bd = NO_D, bi = NO_ID;
        for (int p = 0; p < n_parts; ++p) {
            const MstKey k = gathered[(size_t)p * a.n + v];
            if (key_less(k.d, k.id, bd, bi)) { bd = k.d; bi = k.id; }
        }
        a.vbest[v] = MstKey{bd, bi};
        if (bi != NO_ID) cb_table_min(s_key, s_val, a.comp[v], bd);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CB_SLOTS; i += 256) {
        const int c = s_key[i];
        if (c < 0) continue;
        // look first: most workgroups cannot lower a large component's minimum (a stale look only costs a redundant atomic)
        unsigned long long* slot = &a.cb_d[c];
        if (s_val[i] < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, s_val[i]);
    }
}

// second atomic phase: among the vertices that reach their component's smallest distance, the smallest id
__global__ __launch_bounds__(256) void boruvka_pick_kernel(BoruvkaArgs a)
{
    __shared__ int s_key[CB_SLOTS];
    __shared__ unsigned long long s_val[CB_SLOTS];
    for (int i = threadIdx.x; i < CB_SLOTS; i += 256) {
        s_key[i] = -1;
        s_val[i] = NO_ID;
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) {
        const MstKey k = a.vbest[v];
        if (k.id != NO_ID) {
            const int c = a.comp[v];
            if (k.d == a.cb_d[c]) cb_table_min(s_key, s_val, c, k.id);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CB_SLOTS; i += 256) {
        const int c = s_key[i];
        if (c < 0) continue;
        unsigned long long* slot = &a.cb_id[c];
        if (s_val[i] < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, s_val[i]);
    }
}

// every component root hooks itself to the component at the other end of its edge
__global__ __launch_bounds__(256) void boruvka_hook_kernel(BoruvkaArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n || a.comp[c] != c) return;
    const unsigned long long id = a.cb_id[c];
    if (id == NO_ID) return; // the last component
    const unsigned long long packed = ~id;
    const int x = (int)(packed >> 32), y = (int)(packed & 0xffffffffull);
    if ((unsigned)x >= (unsigned)a.n || (unsigned)y >= (unsigned)a.n) { // keys are caller-supplied (exchanged between ranks)
        a.counters[1] = 1;
        return;
    }
    const int cx = a.comp[x], cy = a.comp[y];
    if ((cx == c) == (cy == c)) { // the edge must leave component c: both ends inside or both outside = keys of another round
        a.counters[1] = 1;
        return;
    }
    const int other = cx == c ? cy : cx;
    a.parent[c] = other;
    // the edge is recorded once: by its only chooser, or by the smaller of two components that chose each other
    const bool mutual = a.cb_id[other] == id;
    if (!mutual || c < other) {
        const int at = atomicAdd(&a.counters[0], 1);
        if (at >= a.n - 1) return; // cannot happen with consistent keys; the host reports the count
        a.edges[at].from = x;
        a.edges[at].to = y;
        a.edges[at].dist = __longlong_as_double((long long)a.cb_d[c]);
    }
}

// two components that chose each other form a 2-cycle: the smaller one becomes the root
__global__ __launch_bounds__(256) void boruvka_uncycle_kernel(BoruvkaArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n || a.comp[c] != c) return;
    const int p = a.parent[c];
    if (p != c && a.parent[p] == c && c < p) a.parent[c] = c;
}

__global__ __launch_bounds__(256) void boruvka_relabel_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    // consistent keys give a forest (2-cycles were cut above): the walk ends at a root within n steps.  Stale or mixed
    // keys can form a longer hooking cycle: bounded, reported through counters[1] (LCSGPU_E_STATE on the host side)
    int r = a.comp[v], steps = 0;
    for (int p = a.parent[r]; p != r; p = a.parent[r]) {
        r = p;
        if (++steps > a.n) {
            a.counters[1] = 1;
            break;
        }
    }
    a.comp_next[v] = r;
}

hipError_t launch_boruvka_init(const BoruvkaArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(boruvka_init_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boruvka_best(const BoruvkaArgs& a, int elem_size, hipStream_t stream)
{
    const dim3 per_vertex((a.n + 255) / 256), threads(256);
    const int rows = a.r1 - a.r0;
    if (rows > 0) {
        const dim3 cols((unsigned)((a.r1 + COLS_PER_WG - 1) / COLS_PER_WG), (unsigned)a.n_chunks); // columns >= r1 - 1 have no row below them here
        const dim3 row_tiles((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG));
        const size_t lds = a.pow_in_lds ? (size_t)a.pow_n * sizeof(double) : 0;
#define MST_PASSES(T, K, P)                                                                         \
    hipLaunchKernelGGL((boruvka_row_kernel<T, K, P>), row_tiles, threads, lds, stream, a);           \
    hipLaunchKernelGGL((boruvka_col_kernel<T, K, P>), cols, threads, lds, stream, a);
#define MST_PASSES_T(T)                                                                             \
    if (a.kind != 1) { MST_PASSES(T, 0, false) }                                                    \
    else if (a.pow_in_lds) { MST_PASSES(T, 1, true) }                                               \
    else { MST_PASSES(T, 1, false) }
        if (elem_size == 2) { MST_PASSES_T(uint16_t) } else { MST_PASSES_T(uint32_t) }
#undef MST_PASSES_T
#undef MST_PASSES
    }
    hipLaunchKernelGGL(boruvka_fold_kernel, per_vertex, threads, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boruvka_fuse_reset(const BoruvkaArgs& a, bool keep, hipStream_t stream)
{
    hipLaunchKernelGGL(boruvka_fuse_reset_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a, keep ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_boruvka_fuse_fold(const BoruvkaArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(boruvka_fuse_fold_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boruvka_merge(const BoruvkaArgs& a, const MstKey* gathered, int n_parts, hipStream_t stream)
{
    const dim3 per_vertex((a.n + 255) / 256), threads(256);
    hipLaunchKernelGGL(boruvka_reset_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_gather_kernel, per_vertex, threads, 0, stream, a, gathered, n_parts);
    hipLaunchKernelGGL(boruvka_pick_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_hook_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_uncycle_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_relabel_kernel, per_vertex, threads, 0, stream, a);
    return hipGetLastError();
}

} // namespace lcsgpu
