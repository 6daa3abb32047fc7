#include "seqset.h"
#include "trees.h"
#include "noinit.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <fstream>
#include <mutex>
#include <unistd.h>
#include <limits>
#include <queue>
#include <set>
#include <random>
#include <stdexcept>
#include <thread>

namespace famsa_host {

GT gt_from_string(const std::string& name)
{
    if (name == "sl") return GT::MST_Prim;
    if (name == "slink") return GT::SLINK;
    if (name == "upgma") return GT::UPGMA;
    if (name == "upgma_modified") return GT::UPGMA_modified;
    if (name == "nj") return GT::NJ;
    if (name == "chained") return GT::chained;
    throw std::runtime_error("Error: Illegal guide tree method.");
}

static inline size_t tri(size_t i, size_t j) // TriangleMatrix::access, reference tree/TreeDefs.h:115-120
{
    return i >= j ? j + i * (i - 1) / 2 : i + j * (j - 1) / 2;
}

static inline uint64_t pack_ids(int a, int b) // ids_to_uint64, reference tree/MSTPrim.h:432-439
{
    if (a < 0 || b < 0) return 0;
    if (a > b) std::swap(a, b);
    return ((uint64_t)a << 32) + (uint64_t)b;
}

// ---------------------------------------------------------------------------------------------
// -gt sl : Prim's MST on the complete graph + MST -> dendrogram
// ---------------------------------------------------------------------------------------------
namespace {

struct Key { // MSTPrim::dist_t = pair<double, uint64_t>, compared lexicographically
    double d;
    uint64_t id;
    bool operator<(const Key& o) const { return d < o.d || (d == o.d && id < o.id); }
};

struct Edge { // MSTPrim::mst_edge_t (reference tree/MSTPrim.h:452-483): dist holds -d
    int from, to, prim_order;
    double dist;
};
inline bool edge_less(const Edge& x, const Edge& y)
{
    if (x.dist != y.dist) return x.dist > y.dist;
    return pack_ids(x.from, x.to) > pack_ids(y.from, y.to);
}

// oriented LCS lookup for Prim: value for (ref = cur, partner = v)
struct PrimLcs {
    const LcsBuf* tri_buf = nullptr; // symmetric case: lower triangle
    const LcsBuf* sq_buf = nullptr;  // orientation-sensitive case: full square
    int n = 0;
    uint32_t operator()(int ref, int partner) const
    {
        if (sq_buf) return (*sq_buf)[(size_t)ref * n + partner];
        return (*tri_buf)[tri(ref, partner)];
    }
};

template <Distance D>
void mst_prim(LcsSource& src, tree_structure& tree)
{
    const int n = src.n();
    std::vector<int> prim_order(n, n);
    std::vector<Edge> edges;
    edges.reserve(n);
    edges.push_back(Edge{0, 0, 0, 0.0}); // the dummy the reference inserts at index 0
    int next_order = 0;
    prim_order[0] = next_order++;

    std::vector<LcsSource::MstEdge> dev_edges;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since_begin = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_edges = 0;
    if (src.prim_edges((int)D, dev_edges, /*triangle_orientation=*/false)) {
        t_edges = since_begin();
        // the engine ran the n-1 relaxation steps on the device; replay the bookkeeping
        for (const auto& e : dev_edges) {
            edges.push_back(Edge{e.from, e.to, next_order, -e.dist});
            if (prim_order[e.from] == n) prim_order[e.from] = next_order++; else prim_order[e.to] = next_order++;
        }
    } else {
        // The source did not build the tree itself (no GPU reducer, or the triangle does not fit its memory).
        // Small sets: all values up front.  Large sets: O(n) memory like the reference -- one ref against the
        // unprocessed vertices per step, exactly MSTPrim::run_view's calculateDistanceRangeSV call
        // (tree/MSTPrim.cpp:478); each step is one engine call, so this is a last resort, not a fast path.
        const bool sensitive = src.orientation_sensitive();
        const double values = sensitive ? (double)n * n : (double)n * (n - 1) / 2;
        const bool streaming = values * (src.wide() ? 4 : 2) > 16e9 || host_test("prim_streaming");
        LcsBuf buf, rowbuf;
        PrimLcs lcs;
        lcs.n = n;
        if (streaming) {
        } else if (sensitive) {
            // ref = the node just added, partner = the candidate (reference MSTPrim.cpp:478-485): the
            // triangle's fixed orientation is not enough, take both orientations
            std::vector<int> all(n);
            for (int i = 0; i < n; ++i) all[i] = i;
            src.rect(all.data(), n, nullptr, n, buf);
            lcs.sq_buf = &buf;
        } else {
            src.triangle(0, n, buf);
            lcs.tri_buf = &buf;
        }
        Transform<double, D> transform;
        std::vector<Key> key(n, Key{std::numeric_limits<double>::max(), 0});
        std::vector<int> alive;
        alive.reserve(n);
        for (int v = 1; v < n; ++v) alive.push_back(v);
        int cur = 0;
        while (!alive.empty()) {
            const uint32_t len_cur = src.length(cur);
            if (streaming) src.rect(&cur, 1, alive.data(), (int)alive.size(), rowbuf); // LCS(ref = cur, partner = alive[p])
            size_t best_pos = 0;
            for (size_t p = 0; p < alive.size(); ++p) {
                const int v = alive[p];
                const double d = transform(streaming ? rowbuf[p] : lcs(cur, v), len_cur, src.length(v));
                if (d <= key[v].d) {
                    const Key s{d, ~pack_ids(cur, v)};
                    if (s < key[v]) key[v] = s;
                }
                if (key[v] < key[alive[best_pos]]) best_pos = p;
            }
            const int best = alive[best_pos];
            const uint64_t packed = ~key[best].id;
            int a = (int)(packed >> 32), b = (int)(packed & 0xffffffffull);
            if (a > b) std::swap(a, b);
            edges.push_back(Edge{a, b, next_order, -key[best].d});
            if (prim_order[a] == n) prim_order[a] = next_order++; else prim_order[b] = next_order++;
            alive[best_pos] = alive.back();
            alive.pop_back();
            cur = best;
        }
    }

    const double t_replay = since_begin();
    // mst_to_dendogram (reference MSTPrim.cpp:784-833): split every range of the Prim order at
    // its heaviest edge, breadth first, node ids handed out from 2n-2 downwards
    std::vector<int> rev(n);
    for (int i = 0; i < n; ++i) rev[prim_order[i]] = i;
    // sparse table of "max" edges (CMaxRangeQueries, reference MSTPrim.h:370-417)
    const int m = (int)edges.size(); // == n
    int levels = 1;
    while ((1 << levels) <= m) ++levels;
    std::vector<std::vector<int>> st(levels);
    st[0].resize(m);
    for (int i = 0; i < m; ++i) st[0][i] = i;
    auto greater = [&](int x, int y) { return edge_less(edges[y], edges[x]); }; // edges[x] > edges[y]
    for (int l = 1; l < levels; ++l) {
        const int len = m - (1 << l) + 1;
        if (len <= 0) { st[l].clear(); continue; }
        st[l].resize(len);
        for (int j = 0; j < len; ++j) {
            const int x = st[l - 1][j], y = st[l - 1][j + (1 << (l - 1))];
            st[l][j] = greater(x, y) ? x : y;
        }
    }
    auto max_element = [&](int begin, int end) {
        int lev = 0;
        while ((1 << (lev + 1)) <= end - begin) ++lev;
        const int x = st[lev][begin], y = st[lev][end - (1 << lev)];
        return greater(x, y) ? x : y;
    };

    tree.assign((size_t)2 * n - 1, node_t(-1, -1));
    struct Range { int id, from, to; };
    std::queue<Range> q;
    int cur_id = 2 * n - 2;
    q.push(Range{cur_id--, 0, n});
    while (!q.empty()) {
        const Range r = q.front();
        q.pop();
        const int split = edges[max_element(r.from + 1, r.to)].prim_order;
        int left, right;
        if (r.from + 1 == split) left = rev[r.from];
        else {
            left = cur_id--;
            q.push(Range{left, std::min(r.from, split), std::max(r.from, split)});
        }
        if (split + 1 == r.to) right = rev[split];
        else {
            right = cur_id--;
            q.push(Range{right, std::min(split, r.to), std::max(split, r.to)});
        }
        tree[r.id] = node_t(left, right);
    }
    if (profile_on())
        fprintf(stderr, "mst_prim (host): the source's edges %.3f s, Prim bookkeeping %.3f s, MST -> dendrogram %.3f s\n", t_edges,
                t_replay - t_edges, since_begin() - t_replay);
}

// ---------------------------------------------------------------------------------------------
// -gt slink : SLINK pointer representation, rows consumed in order
// ---------------------------------------------------------------------------------------------
struct SlinkDist { // slink_dist_t (reference tree/SingleLinkage.h:19-38): by distance, then by LARGER id
    double first;
    uint64_t second;
    bool operator<(const SlinkDist& r) const { return first == r.first ? second > r.second : first < r.first; }
    bool operator<=(const SlinkDist& r) const { return first == r.first ? second >= r.second : first <= r.first; }
};

// SLINK's output from the minimum spanning tree.  With the strict total order both generators put
// on pair distances -- (d ascending, packed ids descending): slink_dist_t (SingleLinkage.h:19-38)
// and MSTPrim's (d, ~pack) keys -- the single-linkage hierarchy is unique and SLINK's pointer
// representation is its canonical encoding: processing the MST edges in that order, the component
// whose largest member `a` is the smaller of the two maxima gets lambda[a] = edge, pi[a] = the
// other maximum.  The distances must be SLINK's: ref = the larger index (the triangle orientation).
struct SlinkEdge { int from, to; double d; };
void slink_from_mst(std::vector<SlinkEdge>& edges, int n, tree_structure& tree)
{
    std::sort(edges.begin(), edges.end(), [](const SlinkEdge& x, const SlinkEdge& y) {
        if (x.d != y.d) return x.d < y.d;
        return pack_ids(x.from, x.to) > pack_ids(y.from, y.to);
    });
    std::vector<int> parent(n), top(n), pi(n);
    for (int i = 0; i < n; ++i) parent[i] = top[i] = pi[i] = i;
    auto find = [&](int x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    };
    std::vector<int> elements;
    elements.reserve(n - 1);
    for (const SlinkEdge& e : edges) {
        const int A = find(e.from), B = find(e.to);
        const int a = std::min(top[A], top[B]), b = std::max(top[A], top[B]);
        pi[a] = b;
        elements.push_back(a);
        parent[A] = B;
        top[B] = b;
    }
    std::vector<int> index(n);
    for (int i = 0; i < n; ++i) index[i] = i;
    for (int i = 0; i < n - 1; ++i) {
        const int j = elements[i];
        const int next = pi[j];
        tree.emplace_back(index[j], index[next]);
        index[next] = n + i;
    }
}

// Prim on the host over triangle-orientation distances (ref = larger index): test path for slink_from_mst
template <Distance D>
void host_prim_triangle(LcsSource& src, std::vector<SlinkEdge>& out)
{
    const int n = src.n();
    LcsBuf buf;
    src.triangle(0, n, buf);
    Transform<double, D> transform;
    std::vector<Key> key(n, Key{std::numeric_limits<double>::max(), 0});
    std::vector<int> alive;
    for (int v = 1; v < n; ++v) alive.push_back(v);
    int cur = 0;
    while (!alive.empty()) {
        size_t best_pos = 0;
        for (size_t p = 0; p < alive.size(); ++p) {
            const int v = alive[p];
            const int hi = std::max(cur, v), lo = std::min(cur, v);
            const double d = transform(buf[tri(hi, lo)], src.length(hi), src.length(lo));
            if (d <= key[v].d) {
                const Key s{d, ~pack_ids(cur, v)};
                if (s < key[v]) key[v] = s;
            }
            if (key[v] < key[alive[best_pos]]) best_pos = p;
        }
        const int best = alive[best_pos];
        const uint64_t packed = ~key[best].id;
        out.push_back(SlinkEdge{(int)(packed >> 32), (int)(packed & 0xffffffffull), key[best].d});
        alive[best_pos] = alive.back();
        alive.pop_back();
        cur = best;
    }
}

template <Distance D>
void slink(LcsSource& src, tree_structure& tree)
{
    const int n = src.n();
    {
        std::vector<LcsSource::MstEdge> dev_edges;
        std::vector<SlinkEdge> edges;
        if (src.prim_edges((int)D, dev_edges, /*triangle_orientation=*/true)) {
            for (const auto& e : dev_edges) edges.push_back(SlinkEdge{e.from, e.to, e.dist});
            slink_from_mst(edges, n, tree);
            return;
        }
        if (host_test("slink_from_mst")) { // test hook: the same conversion with Prim on the host
            host_prim_triangle<D>(src, edges);
            slink_from_mst(edges, n, tree);
            return;
        }
    }
    Transform<double, D> transform;
    std::vector<int> pi(n, 0);
    std::vector<SlinkDist> lambda(n), M(n);
    // rows are fetched in blocks from the engine (ref = row i, partner = column j < i)
    const int block = std::max(1, std::min(n, 4096));
    LcsBuf buf;
    int b0 = 0, b1 = 0;
    for (int i = 0; i < n; ++i) {
        if (i >= b1) {
            b0 = i;
            b1 = std::min(n, i + block);
            src.triangle(b0, b1, buf);
        }
        const size_t base = (size_t)i * (i - 1) / 2 - (size_t)b0 * (b0 > 0 ? b0 - 1 : 0) / 2;
        pi[i] = i;
        lambda[i] = SlinkDist{std::numeric_limits<double>::max(), 0};
        const uint32_t len_i = src.length(i);
        for (int j = 0; j < i; ++j)
            M[j] = SlinkDist{transform(buf[base + j], len_i, src.length(j)), pack_ids(j, i)};
        for (int j = 0; j < i; ++j) {
            const int next = pi[j];
            SlinkDist& x = M[next];
            if (lambda[j] < M[j]) {
                x = std::min(x, M[j]);
            } else {
                x = std::min(x, lambda[j]);
                pi[j] = i;
                lambda[j] = M[j];
            }
        }
        for (int j = 0; j < i; ++j)
            if (lambda[pi[j]] <= lambda[j]) pi[j] = i;
    }
    std::vector<int> elements(n - 1);
    for (int i = 0; i < n - 1; ++i) elements[i] = i;
    std::stable_sort(elements.begin(), elements.end(), [&](int x, int y) { return lambda[x] < lambda[y]; });
    std::vector<int> index(n);
    for (int i = 0; i < n; ++i) index[i] = i;
    for (int i = 0; i < n - 1; ++i) {
        const int j = elements[i];
        const int next = pi[j];
        tree.emplace_back(index[j], index[next]);
        index[next] = n + i;
    }
}

// ---------------------------------------------------------------------------------------------
// float distance triangle (UPGMA, NJ): d[i*(i-1)/2 + j] = Transform<float>(LCS(ref=i, partner=j))
// ---------------------------------------------------------------------------------------------
template <Distance D>
void float_triangle(LcsSource& src, std::vector<float>& dist)
{
    const int n = src.n();
    dist.resize((size_t)n * (n - 1) / 2);
    Transform<float, D> transform;
    const int block = 8192;
    LcsBuf buf;
    for (int r0 = 0; r0 < n; r0 += block) {
        const int r1 = std::min(n, r0 + block);
        src.triangle(r0, r1, buf);
        const size_t off = (size_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2;
        for (int i = std::max(r0, 1); i < r1; ++i) {
            const uint32_t len_i = src.length(i);
            const size_t row = (size_t)i * (i - 1) / 2;
            for (int j = 0; j < i; ++j) dist[row + j] = transform(buf[row + j - off], len_i, src.length(j));
        }
    }
}

// -gt upgma / upgma_modified.  What has to come out is UPGMA::computeTree's tree (reference tree/UPGMA.cpp:114-295): its
// row statistics are deliberately stale (a row's minimum is only ever replaced when the row itself is re-created by a
// merge; nearest[] changes only by the rename "partner of the merge -> merged row") and every tie goes to the smaller row
// index.  That staleness is what this form -- the host twin of csrc/upgma_batch_kernels.hip -- builds on: because a row's key
// never changes while the row lives, the next pick is simply the smallest (key, row) of an ORDERED SET of the live rows
// instead of a scan over all of them, and the live rows are a linked list in index order, so a merge touches only rows
// that still exist.  The float operations and their order are the reference's.
template <bool MODIFIED>
void upgma_tree(std::vector<float>& D, int n, tree_structure& tree)
{
    constexpr float BIG = 1e29f; // UPGMA::BIG_DIST: a row whose minimum is not below it is never picked
    constexpr int NONE = 0x7FFFFFFF;
    auto average = [](float x, float y) -> float {
        if (MODIFIED) return 0.05f * (x + y) + 0.9f * std::min(x, y);
        return (x + y) * 0.5f;
    };
    // every row's first strict minimum over the other rows in ascending index, and the node each row stands for
    std::vector<float> key(n, BIG);
    std::vector<int> partner(n, NONE), node(n), next(n + 1), prev(n + 1);
    for (int x = 0; x < n; ++x) {
        node[x] = x;
        next[x] = x + 1;
        prev[x + 1] = x;
        for (int y = 0; y < n; ++y) {
            if (y == x) continue;
            const float d = D[tri(x, y)];
            if (d < key[x]) { key[x] = d; partner[x] = y; }
        }
    }
    int first = 0; // head of the list of live rows (next[] ends at n)
    std::set<std::pair<float, int>> order;
    for (int x = 0; x < n; ++x)
        if (key[x] < BIG) order.emplace(key[x], x);
    auto unlink = [&](int x) {
        if (x == first) first = next[x];
        else next[prev[x]] = next[x];
        prev[next[x]] = prev[x];
    };
    for (int made = 0; made < n - 1; ++made) {
        if (order.empty() || partner[order.begin()->second] == NONE)
            throw std::runtime_error("UPGMA: no finite nearest neighbour (a pair with LCS 0?) -- the reference's "
                                     "algorithm is undefined for this input");
        const int keep = order.begin()->second, gone = partner[keep]; // the merged cluster keeps the picked row
        order.erase(order.begin());
        if (key[gone] < BIG) order.erase({key[gone], gone});
        unlink(gone);
        float best = BIG;
        int best_row = NONE;
        for (int j = first; j < n; j = next[j]) { // ascending: the first strict minimum is the reference's
            if (j == keep) continue;
            float& cell = D[tri(keep, j)];
            cell = average(cell, D[tri(gone, j)]);
            if (partner[j] == gone) partner[j] = keep;
            if (cell < best) { best = cell; best_row = j; }
        }
        tree.emplace_back(node[keep], node[gone]);
        node[keep] = n + made;
        key[keep] = best;
        partner[keep] = best_row;
        if (best < BIG) order.emplace(best, keep);
    }
}

// ---- the same tree from a SQUARE float matrix: the form the leaves of the FastTree recursion take -------------------------
// A leaf is at most `threshold` (2000) sequences and there are thousands of them, so what counts is the constant per matrix
// element, not the order of growth: the triangle form above walks a linked list and reads down a column (one cache line per
// element) for every row beyond the merged one.  Here the matrix is m x ld floats, symmetric: a merge averages two contiguous
// rows four elements at a time (SSE2, part of the x86-64 baseline -- the same IEEE add / mul / min / div, one rounding each,
// no contraction), rows that are gone are skipped by a lane mask instead of a list, the new row is mirrored into its column by
// plain stores, and the next pick comes from a heap of (key, row) with lazy deletion (a row's key never changes while it
// lives, see above).  Same picks, same float operations in the same order per element, same ties.
#if defined(__SSE2__)
constexpr int UPGMA_SQUARE_MAX = 4096; // 64 MB of floats; above that the triangle form

inline float hmin4(__m128 v)
{
    v = _mm_min_ps(v, _mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 0, 3, 2)));
    v = _mm_min_ps(v, _mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 3, 0, 1)));
    return _mm_cvtss_f32(v);
}
inline __m128 select4(__m128 mask, __m128 a, __m128 b) { return _mm_or_ps(_mm_and_ps(mask, a), _mm_andnot_ps(mask, b)); }

// first strict minimum below BIG of row[j] over the columns whose mask is set, ascending j: (value, column) or (BIG, NONE)
inline void masked_first_min(const float* row, const uint32_t* mask, int ld, float big, float& value, int& column)
{
    const __m128 bigv = _mm_set1_ps(big);
    __m128 best = bigv;
    for (int j = 0; j < ld; j += 4)
        best = _mm_min_ps(best, select4(_mm_castsi128_ps(_mm_loadu_si128((const __m128i*)(mask + j))), _mm_loadu_ps(row + j), bigv));
    value = hmin4(best);
    column = 0x7FFFFFFF;
    if (!(value < big)) { value = big; return; }
    const __m128 want = _mm_set1_ps(value);
    for (int j = 0; j < ld; j += 4) {
        const __m128 c = select4(_mm_castsi128_ps(_mm_loadu_si128((const __m128i*)(mask + j))), _mm_loadu_ps(row + j), bigv);
        const int hit = _mm_movemask_ps(_mm_cmpeq_ps(c, want));
        if (hit) { column = j + __builtin_ctz(hit); return; }
    }
}

// pow(i, 0.75) as float, i <= upto: one table per thread, grown on demand (Transform<float, indel075_div_lcs>::pow075)
inline const float* pow075_table(size_t upto)
{
    static thread_local std::vector<float> table;
    if (upto >= table.size()) {
        const size_t from = table.size();
        table.resize(upto + 1);
        for (size_t i = from; i <= upto; ++i) table[i] = (float)std::pow((double)(uint32_t)i, 0.75);
    }
    return table.data();
}

// out[j] = Transform<float, D>(l[j], len_i, lens[j]), j < count (reference tree/AbstractTreeGenerator.hpp:28-82): the division
// four at a time, the table values gathered one by one; a pair without a common residue gets the reference's FLT_MAX
template <Distance D, class E>
void transform_row(const E* l, uint32_t len_i, const uint32_t* lens, int count, const float* pw, float* out)
{
    static_assert(D != Distance::pairwise_identity, "tree distances only");
    const __m128 none = _mm_set1_ps(zero_lcs_distance<float>());
    int j = 0;
    const __m128i vlen = _mm_set1_epi32((int)len_i);
    for (; j + 4 <= count; j += 4) {
        __m128i li;
        if (sizeof(E) == 2) li = _mm_unpacklo_epi16(_mm_loadl_epi64((const __m128i*)(l + j)), _mm_setzero_si128());
        else li = _mm_loadu_si128((const __m128i*)(l + j));
        const __m128i indel = _mm_sub_epi32(_mm_add_epi32(vlen, _mm_loadu_si128((const __m128i*)(lens + j))), _mm_add_epi32(li, li));
        __m128 num;
        if (D == Distance::indel075_div_lcs) {
            alignas(16) uint32_t ix[4];
            _mm_store_si128((__m128i*)ix, indel);
            num = _mm_set_ps(pw[ix[3]], pw[ix[2]], pw[ix[1]], pw[ix[0]]);
        } else {
            num = _mm_cvtepi32_ps(indel);
        }
        const __m128 q = _mm_div_ps(num, _mm_cvtepi32_ps(li));
        const __m128 is_zero = _mm_castsi128_ps(_mm_cmpeq_epi32(li, _mm_setzero_si128()));
        _mm_storeu_ps(out + j, select4(is_zero, none, q));
    }
    for (; j < count; ++j) { // the tail, element by element
        const uint32_t lj = l[j], indel = len_i + lens[j] - 2 * lj;
        const float num = D == Distance::indel075_div_lcs ? pw[indel] : (float)indel;
        out[j] = lj ? num / (float)lj : zero_lcs_distance<float>();
    }
}

template <bool MODIFIED, Distance D>
void upgma_square(LcsSource& src, tree_structure& tree)
{
    constexpr float BIG = 1e29f;
    constexpr int NONE = 0x7FFFFFFF;
    const int m = src.n();
    int dim = m, ld = (m + 3) & ~3; // rows in the matrix (live or not) and its row pitch; both shrink when the matrix is packed
    std::vector<uint32_t> lens(ld, 0u);
    uint32_t longest = 0;
    for (int i = 0; i < m; ++i) longest = std::max(longest, lens[i] = src.length(i));
    const float* pw = D == Distance::indel075_div_lcs ? pow075_table((size_t)2 * longest) : nullptr;
    // one per pool thread (freed when the recursion's pool ends), as large as its largest recent leaf: given back when a
    // leaf needs less than a quarter of it, so that one 2000-member leaf does not pin 16 MB per thread for a run of small ones
    static thread_local std::vector<float> matrix;
    if (matrix.capacity() > ((size_t)1 << 20) && (size_t)m * ld * 4 < matrix.capacity()) std::vector<float>().swap(matrix);
    if (matrix.size() < (size_t)m * ld) matrix.resize((size_t)m * ld);
    float* const M = matrix.data();
    std::vector<float> key(ld, BIG);
    std::vector<int> partner(ld, NONE), node(m), live(m);

    // 1. the distances, row by row over the lower triangle, and while a row is hot every vertex's first strict minimum in
    //    ascending order of the other vertex: row x settles x against the y < x, then offers itself to every y < x -- so a
    //    vertex sees its own row first and the rows below it in ascending order, as the reference's sweep does
    {
        LcsBuf own;
        const void* view = src.triangle_view();
        bool wide = src.wide();
        if (!view) {
            src.triangle(0, m, own);
            wide = own.wide;
            view = own.wide ? (const void*)own.v32.data() : (const void*)own.v16.data();
        }
        for (int i = 0; i < m; ++i) {
            float* const row = M + (size_t)i * ld;
            std::fill(row + i, row + std::min(ld, i + 4), BIG); // what the sweep below reads past its last column; the rest: the mirror
            if (i == 0) continue;
            const size_t first = (size_t)i * (i - 1) / 2;
            if (wide) transform_row<D>((const uint32_t*)view + first, lens[i], lens.data(), i, pw, row);
            else transform_row<D>((const uint16_t*)view + first, lens[i], lens.data(), i, pw, row);
            __m128 best = _mm_set1_ps(BIG);
            const __m128i vi = _mm_set1_epi32(i);
            for (int j = 0; j < i; j += 4) { // columns i .. : BIG, never below a key
                const __m128 d = _mm_loadu_ps(row + j), k = _mm_loadu_ps(key.data() + j);
                best = _mm_min_ps(best, d);
                const __m128 lt = _mm_cmplt_ps(d, k);
                _mm_storeu_ps(key.data() + j, select4(lt, d, k));
                const __m128i pj = _mm_loadu_si128((const __m128i*)(partner.data() + j)), ltm = _mm_castps_si128(lt);
                _mm_storeu_si128((__m128i*)(partner.data() + j), _mm_or_si128(_mm_and_si128(ltm, vi), _mm_andnot_si128(ltm, pj)));
            }
            const float value = hmin4(best);
            if (value < BIG) {
                const __m128 want = _mm_set1_ps(value);
                for (int j = 0; j < i; j += 4) {
                    const int hit = _mm_movemask_ps(_mm_cmpeq_ps(_mm_loadu_ps(row + j), want));
                    if (hit) { partner[i] = j + __builtin_ctz(hit); break; }
                }
                key[i] = value;
            }
        }
    }
    constexpr int TB = 32;
    for (int i0 = 0; i0 < m; i0 += TB)
        for (int j0 = 0; j0 <= i0; j0 += TB)
            for (int i = i0; i < std::min(m, i0 + TB); ++i)
                for (int j = j0; j < std::min(i, j0 + TB); ++j) M[(size_t)j * ld + i] = M[(size_t)i * ld + j];

    // 2. the order of the picks: a heap of (key, row), entries of rows that left or were re-created are dropped when they surface
    std::vector<uint32_t> mask(ld, 0u);
    struct Entry { float key; int row, node; };
    auto later = [](const Entry& a, const Entry& b) { return a.key > b.key || (a.key == b.key && a.row > b.row); };
    std::vector<Entry> heap;
    heap.reserve((size_t)2 * m);
    for (int x = 0; x < m; ++x) {
        node[x] = live[x] = x;
        mask[x] = ~0u;
        if (key[x] < BIG) heap.push_back(Entry{key[x], x, x});
    }
    std::make_heap(heap.begin(), heap.end(), later);
    std::vector<int> place; // packing: old row -> new row

    // 3. the merges
    const __m128 bigv = _mm_set1_ps(BIG), half = _mm_set1_ps(0.5f), c005 = _mm_set1_ps(0.05f), c09 = _mm_set1_ps(0.9f);
    for (int made = 0; made < m - 1; ++made) {
        if ((int)live.size() * 2 <= dim && dim >= 128) {
            // Half of the rows are gone: pack the live ones (their order, which settles every tie, stays) so that a merge
            // sweeps and mirrors half as much and the matrix drops into the next cache level.  In place: every element moves
            // towards the front, rows and columns in ascending order.
            const int nl = (int)live.size(), nld = (nl + 3) & ~3;
            place.assign(dim, NONE);
            for (int r = 0; r < nl; ++r) place[live[r]] = r;
            for (int r = 0; r < nl; ++r) {
                const float* from = M + (size_t)live[r] * ld;
                float* to = M + (size_t)r * nld;
                for (int c = 0; c < nl; ++c) to[c] = from[live[c]];
            }
            heap.clear();
            for (int r = 0; r < nl; ++r) {
                const int o = live[r];
                key[r] = key[o];
                partner[r] = partner[o] == NONE ? NONE : place[partner[o]];
                node[r] = node[o];
                if (key[r] < BIG) heap.push_back(Entry{key[r], r, node[r]});
            }
            std::make_heap(heap.begin(), heap.end(), later);
            for (int r = 0; r < nld; ++r) {
                mask[r] = r < nl ? ~0u : 0u;
                if (r >= nl) partner[r] = NONE;
            }
            for (int r = 0; r < nl; ++r) live[r] = r;
            dim = nl;
            ld = nld;
        }
        while (!heap.empty() && (mask[heap.front().row] == 0u || node[heap.front().row] != heap.front().node)) {
            std::pop_heap(heap.begin(), heap.end(), later);
            heap.pop_back();
        }
        if (heap.empty() || partner[heap.front().row] == NONE)
            throw std::runtime_error("UPGMA: no finite nearest neighbour (a pair with LCS 0?) -- the reference's "
                                     "algorithm is undefined for this input");
        const int keep = heap.front().row, gone = partner[keep];
        std::pop_heap(heap.begin(), heap.end(), later);
        heap.pop_back();
        mask[gone] = 0u;
        live.erase(std::lower_bound(live.begin(), live.end(), gone));
        mask[keep] = 0u; // not a candidate of its own row
        float* const rk = M + (size_t)keep * ld;
        const float* const rg = M + (size_t)gone * ld;
        __m128 best = bigv;
        for (int j = 0; j < ld; j += 4) {
            const __m128 x = _mm_loadu_ps(rk + j), y = _mm_loadu_ps(rg + j);
            const __m128 v = MODIFIED ? _mm_add_ps(_mm_mul_ps(c005, _mm_add_ps(x, y)), _mm_mul_ps(c09, _mm_min_ps(x, y)))
                                      : _mm_mul_ps(_mm_add_ps(x, y), half);
            _mm_storeu_ps(rk + j, v);
            best = _mm_min_ps(best, select4(_mm_castsi128_ps(_mm_loadu_si128((const __m128i*)(mask.data() + j))), v, bigv));
        }
        float value = hmin4(best);
        int column = NONE;
        if (value < BIG) {
            const __m128 want = _mm_set1_ps(value);
            for (int j = 0; j < ld; j += 4) {
                const __m128 c = select4(_mm_castsi128_ps(_mm_loadu_si128((const __m128i*)(mask.data() + j))), _mm_loadu_ps(rk + j), bigv);
                const int hit = _mm_movemask_ps(_mm_cmpeq_ps(c, want));
                if (hit) { column = j + __builtin_ctz(hit); break; }
            }
        } else {
            value = BIG;
        }
        mask[keep] = ~0u;
        for (const int j : live) M[(size_t)j * ld + keep] = rk[j]; // the mirror (j == keep: the diagonal, never a candidate)
        const __m128i vg = _mm_set1_epi32(gone), vk = _mm_set1_epi32(keep);
        for (int j = 0; j < ld; j += 4) { // rows whose stored neighbour was the row that left now name the merged row
            const __m128i p = _mm_loadu_si128((const __m128i*)(partner.data() + j));
            const __m128i eq = _mm_cmpeq_epi32(p, vg);
            _mm_storeu_si128((__m128i*)(partner.data() + j), _mm_or_si128(_mm_and_si128(eq, vk), _mm_andnot_si128(eq, p)));
        }
        tree.emplace_back(node[keep], node[gone]);
        node[keep] = m + made;
        key[keep] = value;
        partner[keep] = column;
        if (value < BIG) {
            heap.push_back(Entry{value, keep, node[keep]});
            std::push_heap(heap.begin(), heap.end(), later);
        }
    }
}
#else  // no SSE2: the square form is written with its intrinsics; the triangle walk below serves every size
constexpr int UPGMA_SQUARE_MAX = 0;
template <bool MODIFIED, Distance D>
void upgma_square(LcsSource&, tree_structure&) {}
#endif

// -gt nj.  NeighborJoining::computeTree (reference tree/NeighborJoining.cpp:33-118): q(i, j) = (m - 2) d(i, j) - s_i - s_j over
// the live clusters in ascending order of their rows, first strict minimum; the sums s are float accumulations whose order
// is part of the result.  Here the live rows are a linked list in index order (the reference erases from a vector: the same
// order), the sums live per row.
void nj_tree(std::vector<float>& D, int n, tree_structure& tree)
{
    std::vector<float> sum(n, 0.0f);
    std::vector<int> node(n), next(n + 1), prev(n + 1);
    for (int x = 0; x < n; ++x) {
        node[x] = x;
        next[x] = x + 1;
        prev[x + 1] = x;
        for (int y = 0; y < n; ++y)
            if (y != x) sum[x] += D[tri(x, y)];
    }
    int first = 0;
    int made = 0;
    for (int live = n; live > 2; --live, ++made) {
        float q_min = std::numeric_limits<float>::max();
        int a = first, b = first;
        for (int x = first; x < n; x = next[x])
            for (int y = next[x]; y < n; y = next[y]) {
                const float q = (live - 2) * D[tri(x, y)] - sum[x] - sum[y];
                if (q < q_min) { q_min = q; a = x; b = y; }
            }
        const float d_ab = D[tri(a, b)];
        tree.emplace_back(node[a], node[b]);
        node[a] = n + made; // the joined cluster keeps row a; row b leaves
        sum[a] = 0.0f;
        for (int z = first; z < n; z = next[z]) {
            if (z == a || z == b) continue;
            float& d_az = D[tri(a, z)];
            const float d_bz = D[tri(b, z)];
            sum[z] -= d_az + d_bz;
            d_az = (d_az + d_bz - d_ab) / 2;
            sum[z] += d_az;
            sum[a] += d_az;
        }
        if (b == first) first = next[b];
        else next[prev[b]] = next[b];
        prev[next[b]] = prev[b];
    }
    const int x = first, y = next[first];
    tree.emplace_back(node[x], node[y]);
}

// the partial generators: append n-1 internal nodes (local ids)
template <Distance D>
void build_partial_d(LcsSource& src, GT method, tree_structure& tree)
{
    const int n = src.n();
    switch (method) {
    case GT::SLINK: slink<D>(src, tree); break;
    case GT::UPGMA:
    case GT::UPGMA_modified: {
        std::vector<int32_t> left, right;
        if (src.upgma_nodes((int)D, method == GT::UPGMA_modified, left, right)) { // merges ran on the device
            for (int i = 0; i < n - 1; ++i) tree.emplace_back(left[i], right[i]);
            break;
        }
        if (n <= UPGMA_SQUARE_MAX && !host_test("upgma_triangle")) { // the leaves of the FastTree recursion
            if (method == GT::UPGMA) upgma_square<false, D>(src, tree); else upgma_square<true, D>(src, tree);
            break;
        }
        std::vector<float> dist;
        float_triangle<D>(src, dist);
        if (method == GT::UPGMA) upgma_tree<false>(dist, n, tree); else upgma_tree<true>(dist, n, tree);
        break;
    }
    case GT::NJ: {
        std::vector<int32_t> left, right;
        if (src.nj_nodes((int)D, left, right)) { // merges ran on the device
            for (int i = 0; i < n - 1; ++i) tree.emplace_back(left[i], right[i]);
            break;
        }
        std::vector<float> dist;
        float_triangle<D>(src, dist);
        nj_tree(dist, n, tree);
        break;
    }
    default: throw std::runtime_error("not a partial generator");
    }
}

} // namespace

void build_tree_partial(LcsSource& src, GT method, Distance dist, tree_structure& tree)
{
    if (src.n() < 2) return;
    if (dist == Distance::indel_div_lcs) build_partial_d<Distance::indel_div_lcs>(src, method, tree);
    else if (dist == Distance::indel075_div_lcs) build_partial_d<Distance::indel075_div_lcs>(src, method, tree);
    else throw std::runtime_error("Error: Illegal pairwise distance measure.");
}

void build_tree(LcsSource& src, GT method, Distance dist, tree_structure& tree, int /*n_threads*/)
{
    const int n = src.n();
    tree.assign(std::max(n, 0), node_t(-1, -1));
    if (n < 2) return;
    if (method == GT::chained) throw std::runtime_error("Error: Illegal guide tree method."); // no distances: build_tree_chained
    if (method == GT::MST_Prim) {
        if (dist == Distance::indel_div_lcs) mst_prim<Distance::indel_div_lcs>(src, tree);
        else if (dist == Distance::indel075_div_lcs) mst_prim<Distance::indel075_div_lcs>(src, tree);
        else throw std::runtime_error("Error: Illegal pairwise distance measure.");
        return;
    }
    build_tree_partial(src, method, dist, tree);
}

void build_tree_chained(int n, uint32_t seed, tree_structure& tree)
{
    tree.assign(std::max(n, 0), node_t(-1, -1));
    if (n < 2) return;
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::mt19937 g(seed);
    for (int i = 0; i + 1 < n; ++i) { // Fisher-Yates; the draw is det_uniform_int_distribution<int>(i, n - 1) (utils/deterministic_random.h:62-76)
        const uint32_t diff = (uint32_t)(n - 1 - i) + 1, bad = 0xffffffffu / diff;
        uint32_t r;
        do r = (uint32_t)g(); while (r / diff >= bad);
        std::swap(idx[i], idx[i + (int)(r % diff)]);
    }
    tree.emplace_back(idx[0], idx[1]);
    for (int i = 2; i < n; ++i) tree.emplace_back(idx[i], (int)tree.size() - 1);
}

// ---------------------------------------------------------------------------------------------
void tree_from_unique(tree_structure& vt, const std::vector<int>& sorted2unique)
{
    const int n_total = (int)sorted2unique.size();
    const int n_uniques = (int)(vt.size() + 1) / 2; // GuideTree::getSequenceCount
    const int n_dups = n_total - n_uniques;
    if (n_dups == 0) { // nothing was removed: the map is a permutation of the leaves (the identity for a WorkSet)
        bool identity = true;
        for (int i = 0; i < n_total && identity; ++i) identity = sorted2unique[i] == i;
        if (identity) return;
    }
    // the records of every unique sequence, in ascending order: first[u] .. first[u + 1] in `members`
    std::vector<int> first(n_uniques + 1, 0), members(n_total);
    for (int i = 0; i < n_total; ++i) ++first[sorted2unique[i] + 1];
    for (int u = 0; u < n_uniques; ++u) first[u + 1] += first[u];
    {
        std::vector<int> fill(first.begin(), first.end() - 1);
        for (int i = 0; i < n_total; ++i) members[fill[sorted2unique[i]]++] = i;
    }
    std::vector<int> out_ids(n_uniques);
    vt.insert(vt.begin() + n_uniques, (size_t)2 * n_dups, node_t(-1, -1));
    int node_id = n_uniques + n_dups;
    for (int u = 0; u < n_uniques; ++u) {
        const int* o = members.data() + first[u];
        const int count = first[u + 1] - first[u];
        for (int i = 1; i < count; ++i, ++node_id) {
            if (i == 1) vt[node_id] = node_t(o[0], o[1]);
            else vt[node_id] = node_t(o[i], node_id - 1);
        }
        out_ids[u] = count > 1 ? node_id - 1 : o[0];
    }
    for (int i = node_id; i < (int)vt.size(); ++i) {
        node_t& nd = vt[i];
        nd.first = nd.first < n_uniques ? out_ids[nd.first] : nd.first + 2 * n_dups;
        nd.second = nd.second < n_uniques ? out_ids[nd.second] : nd.second + 2 * n_dups;
    }
}

// The text has a closed form: a leaf is `name:1.0`, an inner node `(` left `,` right `):1.0`, the root ends in `);` -- so
// every node's length follows from its children's and every node's position from its parent's.  The generators append a node
// after its children, which turns both into one sweep over the node array each (up for the lengths, down for the
// positions), and then every node writes its own few characters wherever they belong, on all cores: at 10^6 leaves the
// recursive walk spent its time on the cache misses of one thread.  A tree with a child after its parent (none of ours)
// takes the walk.
static std::string newick_by_walk(const tree_structure& tree, const std::vector<const char*>& names)
{
    const int n_leaves = (int)names.size();
    std::string out;
    const int root = (int)tree.size() - 1;
    // explicit stack: (node, state) with state 0 = open, 1 = between children, 2 = close
    std::vector<std::pair<int, int>> st;
    st.emplace_back(root, 0);
    while (!st.empty()) {
        auto& top = st.back();
        const int node = top.first;
        if (node < n_leaves) {
            const char* name = names[node];
            if (*name == '>') ++name;
            out += name;
            out += ":1.0";
            st.pop_back();
            continue;
        }
        if (top.second == 0) {
            out += '(';
            top.second = 1;
            st.emplace_back(tree[node].first, 0);
        } else if (top.second == 1) {
            out += ',';
            top.second = 2;
            st.emplace_back(tree[node].second, 0);
        } else {
            out += node == root ? ");" : "):1.0";
            st.pop_back();
        }
    }
    return out;
}

std::string tree_to_newick(const tree_structure& tree, const std::vector<std::string>& names)
{
    std::vector<const char*> p(names.size());
    for (size_t i = 0; i < names.size(); ++i) p[i] = names[i].c_str();
    return tree_to_newick(tree, p);
}

std::string tree_to_newick(const tree_structure& tree, const std::vector<const char*>& names)
{
    const int n_leaves = (int)names.size(), n_nodes = (int)tree.size();
    if (tree.empty()) return std::string();
    const int root = n_nodes - 1;
    if (root < n_leaves) return newick_by_walk(tree, names); // a single leaf
    for (int i = n_leaves; i < n_nodes; ++i)
        if (tree[i].first < 0 || tree[i].second < 0 || tree[i].first >= i || tree[i].second >= i) return newick_by_walk(tree, names);
    constexpr uint64_t UNSEEN = ~0ull;
    // (3 x 10^6 leaves: 96 MB of tables and 75 MB of text.  Their first touches -- a page fault each, on one thread, most of this
    //  function's 0.18 s -- are spread over the threads: the tables are filled where that can be done side by side, the text's
    //  pages are touched before the string is sized.)
    const int n_threads = std::max(1, default_host_threads());
    const int slices = std::max(1, std::min(n_threads, n_nodes / 65536));
    auto side_by_side = [&](auto&& fn) {
        std::vector<std::thread> team;
        for (int t = 1; t < slices; ++t) team.emplace_back(fn, t);
        fn(0);
        for (auto& w : team) w.join();
    };
    std::vector<uint64_t, NoInit<uint64_t>> size(n_nodes), at(n_nodes);
    auto leaf_name = [&](int v, size_t& len) {
        const char* p = names[v];
        if (*p == '>') ++p;
        len = strlen(p); // as the walk appends it: up to the first NUL
        return p;
    };
    side_by_side([&](int t) {
        const int i0 = (int)((int64_t)n_nodes * t / slices), i1 = (int)((int64_t)n_nodes * (t + 1) / slices);
        for (int i = i0; i < i1; ++i) {
            at[i] = UNSEEN;
            size[i] = 0;
            if (i < n_leaves) {
                size_t len;
                leaf_name(i, len);
                size[i] = len + 4;
            }
        }
    });
    for (int i = n_leaves; i < n_nodes; ++i) size[i] = size[tree[i].first] + size[tree[i].second] + (i == root ? 4 : 7);
    at[root] = 0;
    for (int i = root; i >= n_leaves; --i) {
        if (at[i] == UNSEEN) continue; // not part of the root's tree
        at[tree[i].first] = at[i] + 1;
        at[tree[i].second] = at[i] + 1 + size[tree[i].first] + 1;
    }
    std::string out;
    out.reserve(size[root]);
    if (slices > 1) { // one byte per page of the reserved block (the sizing below writes every byte anyway)
        char* const base = &out[0];
        const size_t total = out.capacity();
        side_by_side([&](int t) {
            for (size_t b = total * (size_t)t / slices, e = total * (size_t)(t + 1) / slices; b < e; b += 4096) ((volatile char*)base)[b] = 0;
        });
    }
    out.resize(size[root], '\0');
    char* const o = &out[0];
    std::vector<std::thread> workers;
    auto fill = [&](int t) {
        const int i0 = (int)((int64_t)n_nodes * t / slices), i1 = (int)((int64_t)n_nodes * (t + 1) / slices);
        for (int i = i0; i < i1; ++i) {
            if (at[i] == UNSEEN) continue;
            if (i < n_leaves) {
                size_t len;
                const char* p = leaf_name(i, len);
                memcpy(o + at[i], p, len);
                memcpy(o + at[i] + len, ":1.0", 4);
            } else {
                o[at[i]] = '(';
                o[at[i] + size[tree[i].first] + 1] = ',';
                if (i == root) memcpy(o + at[i] + size[i] - 2, ");", 2);
                else memcpy(o + at[i] + size[i] - 5, "):1.0", 5);
            }
        }
    };
    for (int t = 1; t < slices; ++t) workers.emplace_back(fill, t);
    fill(0);
    for (auto& w : workers) w.join();
    return out;
}

// ---------------------------------------------------------------------------------------------
// double -> int64 as the reference's x86-64 build converts it (cvttsd2si: out-of-range values,
// e.g. the lcs == 0 distance, become INT64_MIN); in range it is plain truncation
static inline int64_t trunc_i64(double v)
{
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)v;
}

static int put_u64(uint64_t v, char* out)
{
    char tmp[24];
    int n = 0, k = 0;
    if (v == 0) tmp[n++] = '0';
    for (; v; v /= 10) tmp[n++] = (char)('0' + v % 10);
    while (n) out[k++] = tmp[--n];
    return k;
}

int format_distance(double val, char* out)
{
    // NumericConversions::Double2PChar(val, 6, out) (reference utils/conversion.h:109-119)
    const int64_t a = trunc_i64(val);
    const int64_t b = trunc_i64((1.0 + (val - (double)a)) * 1000000.0 + 0.5);
    const int r1 = put_u64((uint64_t)a, out);
    const int r2 = put_u64((uint64_t)b, out + r1);
    out[r1] = '.';
    return r1 + r2;
}

// ---------------------------------------------------------------------------------------------
// A team of threads that puts finished blocks of text into the file at known offsets (pwrite), so that writing block k
// overlaps the device's work on block k + 1 and the page-cache copy is not one thread's job.
namespace {
class BlockWriter {
public:
    BlockWriter(int fd, int n_threads) : fd_(fd)
    {
        for (int t = 0; t < n_threads; ++t) team_.emplace_back([this] { work(); });
    }
    ~BlockWriter()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : team_) t.join();
    }
    // returns a ticket for wait(); the bytes must stay valid until then
    size_t write(const char* p, uint64_t len, uint64_t offset)
    {
        const uint64_t chunk = 2u << 20;
        std::lock_guard<std::mutex> lk(mu_);
        const size_t ticket = left_.size();
        const uint64_t pieces = (len + chunk - 1) / chunk;
        left_.push_back((size_t)pieces);
        for (uint64_t k = 0; k < pieces; ++k) queue_.push_back({p + k * chunk, std::min(chunk, len - k * chunk), offset + k * chunk, ticket});
        cv_.notify_all();
        return ticket;
    }
    void wait(size_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return left_[ticket] == 0; });
        if (!error_.empty()) throw std::runtime_error(error_);
    }

private:
    struct Piece { const char* p; uint64_t len, offset; size_t ticket; };
    void work()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (queue_.empty()) return;
            Piece w = queue_.front();
            queue_.pop_front();
            lk.unlock();
            std::string err;
            while (w.len) {
                const ssize_t k = pwrite(fd_, w.p, (size_t)w.len, (off_t)w.offset);
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) {
                    err = std::string("writing the distance file failed: ") + (k < 0 ? strerror(errno) : "no progress (disk full?)");
                    break;
                }
                w.p += k;
                w.len -= (uint64_t)k;
                w.offset += (uint64_t)k;
            }
            lk.lock();
            if (!err.empty() && error_.empty()) error_ = err;
            if (--left_[w.ticket] == 0) done_.notify_all();
        }
    }
    int fd_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<Piece> queue_;
    std::vector<size_t> left_;
    std::vector<std::thread> team_;
    std::string error_;
    bool stop_ = false;
};
} // namespace

// -dist_export with a source that makes the text itself (the GPU engine: lcsgpu_dist_text_*): this side only decides the
// row blocks, keeps `units` of them in flight and puts every finished block into the file while the next ones are
// computed, formatted and copied.  What DistanceCalculator::run (reference tree/DistanceCalculator.cpp:10-120) does with a
// queue of row vectors and one writing thread.  Returns false if the source does not offer it.
static bool write_csv_text(LcsSource& src, const std::vector<std::string>& ids, int distance_kind, bool square, bool pid,
                           const std::string& path)
{
    const int n = src.n();
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now();
    const int units = src.text_begin(ids, distance_kind, square, pid);
    if (units <= 0) return false;
    struct End {
        LcsSource& s;
        double t0;
        ~End()
        {
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            try { s.text_end(); } catch (...) {}
            if (profile_on())
                fprintf(stderr, "dist_export.text: text_end %.3f s; stage %.3f s\n",
                        std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t, t - t0);
        }
    } end{src, t_enter};
    const double t_begun = now();
    const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0666);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct Close {
        int fd;
        ~Close() { if (fd >= 0) close(fd); }
    } closer{fd};
    uint64_t offset = 0;
    std::string header;
    if (square) {
        for (const auto& id : ids) {
            header += ',';
            header += id.c_str() + 1;
        }
        header += '\n';
    }
    // row blocks of about equal text: ~9 bytes per value; at most n / 8 rows, so that the part of a triangle block's
    // rectangle that lies beyond the diagonal stays small
    std::vector<double> row_bytes((size_t)n);
    double total = 0;
    for (int i = 0; i < n; ++i) total += row_bytes[(size_t)i] = (double)ids[(size_t)i].size() + 9.2 * (square ? n : i);
    const double mb = 1024.0 * 1024.0;
    const double target = std::min(std::max(total / 32, 4 * mb), (double)host_test_int("text_block_mb", 32) * mb);
    const int max_rows = std::min(32768, std::max(512, n / 8));
    std::vector<std::pair<int, int>> blocks;
    for (int r0 = 0; r0 < n;) {
        int r1 = r0;
        double acc = 0;
        while (r1 < n && r1 - r0 < max_rows && (r1 == r0 || acc + row_bytes[(size_t)r1] <= target)) acc += row_bytes[(size_t)r1++];
        blocks.emplace_back(r0, r1);
        r0 = r1;
    }
    const int nb = (int)blocks.size();
    int next = 0;
    for (int u = 0; u < units && next < nb; ++u, ++next) src.text_submit(u, blocks[(size_t)next].first, blocks[(size_t)next].second);
    BlockWriter writer(fd, std::max(1, std::min(default_host_threads(), host_test_int("text_writers", 8))));
    if (!header.empty()) {
        writer.wait(writer.write(header.data(), header.size(), 0));
        offset = header.size();
    }
    std::vector<size_t> ticket((size_t)nb);
    double t_wait = 0, t_write = 0, t_submit = 0, t_first = 0, t_queue = 0;
    const double t_begin = now();
    for (int k = 0; k < nb; ++k) {
        const char* text = nullptr;
        uint64_t bytes = 0;
        double t0 = now();
        src.text_wait(k % units, text, bytes); // block k is in host memory (its copy ran beside block k - 1's write)
        t_wait += now() - t0;
        if (k == 0) t_first = now() - t_begin;
        t0 = now();
        ticket[(size_t)k] = writer.write(text, bytes, offset);
        t_queue += now() - t0;
        offset += bytes;
        if (k >= 1) { // block k - 1 is in the file: its unit takes the next block
            t0 = now();
            writer.wait(ticket[(size_t)k - 1]);
            t_write += now() - t0;
            if (next < nb) {
                t0 = now();
                src.text_submit((k - 1) % units, blocks[(size_t)next].first, blocks[(size_t)next].second);
                t_submit += now() - t0;
                ++next;
            }
        }
    }
    double t0 = now();
    if (nb > 0) writer.wait(ticket[(size_t)nb - 1]);
    t_write += now() - t0;
    closer.fd = -1;
    if (close(fd) != 0) throw std::runtime_error("writing " + path + " failed (disk full?)");
    if (profile_on())
        fprintf(stderr, "dist_export.text: text_begin %.3f s, file + first submits + writer team %.3f s\n", t_begun - t_enter, t_begin - t_begun);
    if (profile_on())
        fprintf(stderr, "dist_export.text: %d blocks on %d units, %.1f MB; first block ready after %.3f s; main thread waited %.3f s for "
                        "blocks (device + copy), %.3f s for the writers, %.3f s in submits, %.3f s queueing writes; loop %.3f s\n",
                nb, units, offset / 1e6, t_first, t_wait, t_write, t_submit, t_queue, now() - t_begin);
    return true;
}

template <Distance D>
static void write_csv_d(LcsSource& src, const std::vector<std::string>& ids, bool square, bool pid,
                        const std::string& path)
{
    const int n = src.n();
    std::ofstream ofs(path, std::ios::binary);
    if (!ofs.good()) throw std::runtime_error("cannot open " + path);
    if (square) {
        for (const auto& id : ids) ofs << ',' << (id.c_str() + 1);
        ofs << std::endl;
    }
    // LCS rows come from the engine in blocks; the rows of a block are turned into text by all host
    // threads (the reference formats inside its row workers, DistanceCalculator.cpp:28-82) and written in order
    const int n_threads = std::max(1, default_host_threads());
    const size_t value_text = 48; // upper bound of one formatted value: a saturated integer part has 20 digits
    const size_t row_text = 64 + (size_t)n * value_text;
    // The set is exported in the order it was read, i.e. NOT sorted by length, and a wave of the LCS kernels walks to the
    // longest of its 64 partners (-15 % of the kernel's rate on a family set in input order, profiles/bench_workloads_r05.txt).
    // So a block of rows is asked for as a rectangle whose COLUMNS are in length order -- for the triangle: the columns
    // j < r1 -- and the text is written from it column by original column.  Blocks of n / 32 rows keep what the rectangle
    // computes beyond the triangle (the part j >= i of the block's rows) at ~3 %.
    const int block = (int)std::max<size_t>(1, std::min<size_t>({(size_t)n, (size_t)2048, ((size_t)256 << 20) / row_text,
                                                                 square ? (size_t)2048 : std::max<size_t>(256, (size_t)n / 32)}));
    std::vector<int> by_length(n);
    for (int i = 0; i < n; ++i) by_length[i] = i;
    if (!host_test("csv_input_order")) // (FAMSA_HOST_TEST: the columns as they were read -- what the length order is measured against)
        std::stable_sort(by_length.begin(), by_length.end(), [&](int a, int b) { return src.length(a) > src.length(b); });
    LcsBuf buf;
    std::vector<int> refs, cols, where(n, -1); // where[j] = column of sequence j in the block's rectangle
    std::vector<std::vector<char>> text(n_threads);
    std::vector<std::vector<size_t>> row_end(n_threads);
    for (int r0 = 0; r0 < n; r0 += block) {
        const int r1 = std::min(n, r0 + block);
        refs.resize(r1 - r0);
        for (int i = r0; i < r1; ++i) refs[i - r0] = i;
        const int col_limit = square ? n : r1 - 1; // the triangle's last row needs the columns j < r1 - 1
        if ((int)cols.size() != col_limit || !square) {
            cols.clear();
            for (int j : by_length)
                if (j < col_limit) {
                    where[j] = (int)cols.size();
                    cols.push_back(j);
                }
        }
        const int n_cols = (int)cols.size();
        if (n_cols > 0) src.rect(refs.data(), r1 - r0, cols.data(), n_cols, buf);
        // thread w formats the rows r0 + w, r0 + w + T, ... (triangle rows grow: interleaving balances them)
        const int T = std::min(n_threads, r1 - r0);
        auto format_rows = [&](int w) {
            Transform<double, D> t_dist;
            Transform<float, Distance::pairwise_identity> t_pid;
            std::vector<char>& out = text[w];
            std::vector<size_t>& ends = row_end[w];
            ends.clear();
            size_t need = 0;
            for (int i = r0 + w; i < r1; i += T) need += 64 + ids[i].size() + (size_t)(square ? n : i) * value_text;
            if (out.size() < need) out.resize(need);
            char* p = out.data();
            for (int i = r0 + w; i < r1; i += T) {
                p += sprintf(p, "%s,", ids[i].c_str() + 1);
                const int cols = square ? n : i;
                const uint32_t len_i = src.length(i);
                for (int j = 0; j < cols; ++j) {
                    const uint32_t l = buf[(size_t)(i - r0) * n_cols + where[j]];
                    // the reference stores both kinds as float before printing (DistanceCalculator.cpp:44-76)
                    const float v = pid ? t_pid(l, len_i, src.length(j)) : (float)t_dist(l, len_i, src.length(j));
                    p += format_distance((double)v, p);
                    *p++ = ',';
                }
                --p;
                *p++ = '\n';
                ends.push_back((size_t)(p - out.data()));
            }
        };
        std::vector<std::thread> workers;
        for (int w = 1; w < T; ++w) workers.emplace_back(format_rows, w);
        format_rows(0);
        for (auto& t : workers) t.join();
        std::vector<size_t> next(T, 0), begin(T, 0);
        for (int i = r0; i < r1; ++i) {
            const int w = (i - r0) % T;
            const size_t e = row_end[w][next[w]++];
            ofs.write(text[w].data() + begin[w], (std::streamsize)(e - begin[w]));
            begin[w] = e;
        }
        if (ofs.fail()) throw std::runtime_error("writing " + path + " failed (disk full?)");
    }
    ofs.close();
    if (ofs.fail()) throw std::runtime_error("writing " + path + " failed (disk full?)");
}

void write_distance_csv(LcsSource& src, const std::vector<std::string>& ids, Distance dist, bool square, bool pid,
                        const std::string& path)
{
    if (write_csv_text(src, ids, (int)dist, square, pid, path)) return;
    // a source without a formatter of its own (tests feed a matrix): format here
    if (dist == Distance::indel_div_lcs) write_csv_d<Distance::indel_div_lcs>(src, ids, square, pid, path);
    else write_csv_d<Distance::indel075_div_lcs>(src, ids, square, pid, path);
}

} // namespace famsa_host
