// fasttree.cpp -- MedoidTree / PartTree guide-tree heuristic over GPU-computed LCS lengths.
// Restates the control flow of the reference's FastTree (tree/FastTree.cpp:56-436) and its CLARANS
// k-medoids (tree/Clustering.cpp:17-305) with the deterministic random helpers of
// utils/deterministic_random.h, so that seeds, assignments and therefore the tree are identical.
// All LCS values come from the LcsSource in three call shapes: one ref x all (row of the longest
// sequence), the sample triangle, and seeds x all rectangles (FastTree.cpp:309-324, 347, 385, 415).
#include <algorithm>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <stdexcept>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

#include "trees.h"

namespace famsa_host {
namespace {

// coarse phase timers, printed when LCSGPU_PROFILE is set
struct PhaseTimers { // printed and zeroed by build_tree_fast
    double lcs = 0, clarans = 0, partial = 0, assign = 0;
} g_phase;
std::mutex g_phase_mu;
// LCSGPU_PROFILE: how many threads are in which phase over the time of the run, in steps of 50 ms (the recursion's timeline)
struct Timeline {
    enum { LCS, CLARANS, PARTIAL, ASSIGN, IDLE, CPU_WAIT, N };
    struct Ev { double t; int8_t phase, delta; };
    std::vector<Ev> ev;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void note(int phase, int delta)
    {
        if (!profile_on()) return;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(g_phase_mu);
        ev.push_back(Ev{t, (int8_t)phase, (int8_t)delta});
    }
    void dump()
    {
        std::lock_guard<std::mutex> lk(g_phase_mu);
        if (ev.empty()) return;
        std::sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.t < b.t; });
        const double step = 0.05, begin = ev.front().t;
        double cur[N] = {0}, acc[N] = {0};
        double t_prev = begin, edge = begin + step;
        fprintf(stderr, "fasttree.timeline (threads per phase, averages over %.0f ms):  t      lcs clarans partial assign  idle cpu_wait\n", 1e3 * step);
        auto flush = [&](double upto) {
            for (int p = 0; p < N; ++p) acc[p] += cur[p] * (upto - t_prev);
            t_prev = upto;
        };
        for (const Ev& e : ev) {
            while (e.t >= edge) {
                flush(edge);
                fprintf(stderr, "fasttree.timeline %6.2f  %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", edge - begin, acc[0] / step, acc[1] / step, acc[2] / step,
                        acc[3] / step, acc[4] / step, acc[5] / step);
                for (int p = 0; p < N; ++p) acc[p] = 0;
                edge += step;
            }
            flush(e.t);
            cur[e.phase] += e.delta;
        }
        ev.clear();
    }
} g_timeline;
struct Scope { // thread-seconds, summed over the worker threads
    double& acc;
    int phase;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    Scope(double& a, int ph) : acc(a), phase(ph) { g_timeline.note(phase, +1); }
    ~Scope()
    {
        g_timeline.note(phase, -1);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(g_phase_mu);
        acc += dt;
    }
};

// The pool below runs more threads than the host grants cores: a thread that waits for the GPU (an LCS request, a
// CLARANS search, a seed assignment) costs no core, and the more searches are in flight the wider the engine's
// CLARANS batches are.  What IS bounded by the cores -- leaf trees, the loops over ids -- runs under one of
// `n_cpu` slots: a thread holds a slot while it works and gives it back for the time it waits for the GPU or for
// its sub-tasks.
class CpuSlots {
public:
    void reset(int n)
    {
        std::lock_guard<std::mutex> lk(mu_);
        free_ = n;
        enabled_ = n > 0;
    }
    void acquire()
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (!enabled_) return;
        if (free_ <= 0) {
            note_cpu_wait(+1);
            cv_.wait(lk, [this] { return free_ > 0; });
            note_cpu_wait(-1);
        }
        --free_;
    }
    void release()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!enabled_) return;
            ++free_;
        }
        cv_.notify_one();
    }

private:
    static void note_cpu_wait(int delta);
    std::mutex mu_;
    std::condition_variable cv_;
    int free_ = 0;
    bool enabled_ = false;
} g_cpu;
void CpuSlots::note_cpu_wait(int delta) { g_timeline.note(Timeline::CPU_WAIT, delta); }
// How many batched leaf requests are with the engine at a time: each takes a lane of the engine, and a lane brings its own
// staging and result buffers (tens of MB of pinned and device memory, allocated -- slowly -- when first needed and when
// outgrown).  Three requests in flight keep the GPU's queue filled; thirty paid for thirty sets of buffers
// (3 x 10^6 sequences: 20-45 ms of allocations per request).
class Slots {
public:
    explicit Slots(int n) : free_(n) {}
    void acquire()
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return free_ > 0; });
        --free_;
    }
    void release()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++free_;
        }
        cv_.notify_one();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int free_;
};
Slots g_leaf_requests(3);

struct OffCpu { // around a wait for the GPU
    OffCpu() { g_cpu.release(); }
    ~OffCpu() { g_cpu.acquire(); }
};

inline size_t tri(size_t i, size_t j) { return i >= j ? j + i * (i - 1) / 2 : i + j * (j - 1) / 2; }

// det_uniform_int_distribution<IntType>::operator() (reference utils/deterministic_random.h:62-76).
// The parameters travel as pair<int,int>; the arithmetic is done in the unsigned twin of IntType.
template <class UInt, class Gen>
inline long long det_uniform(Gen& g, int lo, int hi)
{
    const UInt diff = (UInt)hi - (UInt)lo + 1;
    if (diff == 0) return (long long)g();
    const UInt bad = std::numeric_limits<UInt>::max() / diff;
    for (;;) {
        const UInt r = (UInt)g();
        if (r / diff < bad) return (long long)((r % diff) + (UInt)lo);
    }
}

// partial_shuffle (deterministic_random.h:113-127); the distribution there is over
// iterator difference_type (64-bit)
template <class Gen>
void partial_shuffle(int* first, int* middle, int* last, Gen& g)
{
    const long n = middle - first, N = last - first - 1;
    for (long i = 0; i < n; ++i) std::swap(first[i], first[det_uniform<unsigned long>(g, (int)i, (int)N)]);
}

// ---- CLARANS k-medoids on the host: the form the device search has (csrc/clarans_kernels.hip), run serially -----------
// Used where there is no device search: a matrix source without a GPU (the CPU suite), more than 1024 medoids, or on
// request (FAMSA_CLARANS_HOST=1, the checker of the device search).  What it must reproduce is CLARANS::operator()
// (reference tree/Clustering.cpp:17-305): the same draws, the same float additions in the same order, the same
// comparison directions -- the sign of a delta decides the search.  How it is organised is this engine's own, and the
// same as on the device: all state is kept per candidate POSITION (pos < k: the medoid in slot pos; pos >= k: a
// non-medoid), the member-to-medoid distances as a slot-major matrix that is kept in step with the swaps, a step is
// EVALUATED (every slot's delta) separately from being APPLIED, and the running cost is a sequential sum over a log
// of addends in position order.
class HostClarans {
public:
    HostClarans(const float* D, int n, int k, int n_fixed) : D_(D), n_(n), k_(k), fixed_(n_fixed), member_(n), st_(n), to_slot_((size_t)k * n), delta_(k) {}

    // one local search from the candidate order `start`; returns its final cost, `member_[0..k)` are its medoids
    template <class Gen>
    float search(const std::vector<int>& start, int steps_without_accept, Gen& draw_position)
    {
        member_ = start;
        float cost = 0.0f;
        for (int pos = k_; pos < n_; ++pos) { // Clustering.cpp:49-79: every non-medoid's two nearest slots, the initial cost
            for (int slot = 0; slot < k_; ++slot) to_slot_[(size_t)slot * n_ + pos] = D_[tri(member_[slot], member_[pos])];
            st_[pos] = two_nearest(pos, -1, 0.0f);
            cost += st_[pos].dn;
        }
        // cpp:86-89, 236: the search ends after `corrected` steps in a row without an accept -- one fewer once a step has
        // been accepted (the reference resets its loop counter to 0 inside the loop, whose increment makes it 1)
        int allowed = steps_without_accept;
        for (int quiet = 0; quiet < allowed;) {
            ++quiet;
            const int xx = (int)det_uniform<unsigned int>(draw_position, k_, n_ - 1);
            const int slot = evaluate(xx);
            if (delta_[slot] < 0.0f) {
                cost = apply(xx, slot, cost);
                quiet = 0;
                allowed = steps_without_accept - 1;
            }
        }
        return cost;
    }
    const std::vector<int>& members() const { return member_; }

private:
    struct Two { float dn, ds; int an, as; }; // distance to / slot of the nearest and the second nearest medoid

    // first and second minimum over the slots in ascending order (cpp:262-305); slot `swap_slot`, if any, at distance d_swap
    Two two_nearest(int pos, int swap_slot, float d_swap) const
    {
        Two t{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), -1, -1};
        for (int slot = 0; slot < k_; ++slot) {
            const float d = slot == swap_slot ? d_swap : to_slot_[(size_t)slot * n_ + pos];
            if (d < t.dn) { t.ds = t.dn; t.as = t.an; t.dn = d; t.an = slot; }
            else if (d < t.ds) { t.ds = d; t.as = slot; }
        }
        return t;
    }

    // every slot's cost change if its medoid were replaced by the member at xx (cpp:93-122): a non-medoid adds
    // min(d, second) - nearest to its own slot, and d - nearest to every other slot when that is negative; position
    // order, one accumulator per slot.  Returns the first minimum over the free slots.
    int evaluate(int xx)
    {
        std::fill(delta_.begin(), delta_.end(), 0.0f);
        const int x = member_[xx];
        for (int pos = k_; pos < n_; ++pos) {
            if (pos == xx) continue;
            const float d = D_[tri(x, member_[pos])];
            const Two& t = st_[pos];
            const float own = std::min(d, t.ds) - t.dn, other = d - t.dn;
            if (other < 0.0f) {
                for (int slot = 0; slot < k_; ++slot) delta_[slot] += slot == t.an ? own : other;
            } else
                delta_[t.an] += own;
        }
        return (int)(std::min_element(delta_.begin() + fixed_, delta_.end()) - delta_.begin());
    }

    // the swap (cpp:124-238): the member at xx becomes the medoid of `slot`, the old medoid moves to xx; every
    // non-medoid keeps, moves or re-derives its two nearest slots; the cost takes the addends in position order
    float apply(int xx, int slot, float cost)
    {
        const int x = member_[xx], old_medoid = member_[slot];
        cost -= st_[xx].dn; // the new medoid leaves the sum first
        member_[slot] = x;
        member_[xx] = old_medoid;
        for (int pos = k_; pos < n_; ++pos) {
            if (pos == xx) { // the replaced medoid: distances to the new medoid set, a fresh assignment
                for (int s2 = 0; s2 < k_; ++s2) to_slot_[(size_t)s2 * n_ + pos] = D_[tri(member_[s2], old_medoid)];
                st_[pos] = two_nearest(pos, -1, 0.0f);
                cost += st_[pos].dn;
                continue;
            }
            const float d = D_[tri(x, member_[pos])];
            Two& t = st_[pos];
            if (t.an == slot) { // its medoid is the one that left
                if (d < t.ds) {
                    cost += d - t.dn;
                    t.dn = d;
                } else {
                    cost += t.ds - t.dn;
                    t = two_nearest(pos, slot, d);
                }
            } else if (d < t.dn) {
                cost += d - t.dn;
                t = Two{d, t.dn, slot, t.an};
            } else if (t.as != slot && d < t.ds) {
                t.ds = d;
                t.as = slot;
            } else
                t = two_nearest(pos, slot, d);
            to_slot_[(size_t)slot * n_ + pos] = d;
        }
        return cost;
    }

    const float* D_;
    int n_, k_, fixed_;
    std::vector<int> member_;     // by position
    std::vector<Two> st_;         // by position (only positions >= k are used)
    std::vector<float> to_slot_;  // [slot][position]: distance of the member at a position to the medoid in a slot
    std::vector<float> delta_;
};

// CLARANS::operator(): `num_local` local searches, each from a fresh partial shuffle of the candidate order (the two
// generators keep running across the searches), the medoids of the cheapest one (ties: the earlier search)
struct Clarans {
    float explore_fraction;
    int num_local;

    void operator()(const float* D, int n_elems, int n_medoids, int n_fixed, int* medoids) const
    {
        // cpp:21-29: how many steps without an accept end a local search
        const int n_swaps = (n_elems - n_medoids) * n_medoids;
        const int max_neighbor = n_swaps < 250 ? n_swaps : std::max((int)(explore_fraction * n_swaps), 250);
        const int corrected = max_neighbor / n_medoids;
        std::vector<int> order(n_elems);
        std::iota(order.begin(), order.end(), 0);
        std::mt19937 shuffle_gen, position_gen;
        HostClarans hc(D, n_elems, n_medoids, n_fixed);
        float best = std::numeric_limits<float>::max();
        for (int iter = 0; iter < num_local; ++iter) {
            partial_shuffle(order.data() + n_fixed, order.data() + n_elems, order.data() + n_elems, shuffle_gen);
            const float cost = hc.search(order, corrected, position_gen);
            order = hc.members(); // the next search shuffles the order this one ended in
            if (cost < best) {
                best = cost;
                std::copy_n(order.begin(), n_medoids, medoids);
            }
        }
    }
};

// ---- a view of a subset of the parent source, local ids 0..m-1 --------------------------------
class SubsetSource : public LcsSource {
public:
    SubsetSource(LcsSource& parent, const std::vector<int>& ids) : p_(parent), ids_(ids) {}
    int n() const override { return (int)ids_.size(); }
    uint32_t length(int i) const override { return p_.length(ids_[i]); }
    bool orientation_sensitive() const override { return p_.orientation_sensitive(); }
    bool wide() const override { return p_.wide(); }
    void triangle(int r0, int r1, LcsBuf& out) override
    {
        const size_t off = (size_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2;
        out.resize((size_t)r1 * (r1 - 1) / 2 - off, p_.wide());
        if (r1 <= 1 || r1 <= r0) return;
        if (r0 == 0) { // the whole subset: the engine's triangle over an id list
            p_.triangle_ids(ids_.data(), r1, out);
            return;
        }
        LcsBuf rect;
        const int cols = r1 - 1;
        p_.rect(ids_.data() + r0, r1 - r0, ids_.data(), cols, rect); // ref = row, partner = column
        for (int i = std::max(r0, 1); i < r1; ++i)
            for (int j = 0; j < i; ++j) {
                const uint32_t v = rect[(size_t)(i - r0) * cols + j];
                const size_t k = (size_t)i * (i - 1) / 2 + j - off;
                if (out.wide) out.v32[k] = v; else out.v16[k] = (uint16_t)v;
            }
    }
    void rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out) override
    {
        std::vector<int> r(n_refs), c(n_cols);
        for (int i = 0; i < n_refs; ++i) r[i] = ids_[refs[i]];
        for (int i = 0; i < n_cols; ++i) c[i] = ids_[cols ? cols[i] : i];
        p_.rect(r.data(), n_refs, c.data(), n_cols, out);
    }
    bool assign_seeds(const int* seeds, int n_seeds, const int* cols, int n_cols, int kind, int first_k, float* dist,
                      int* assign) override
    {
        std::vector<int> sg(n_seeds), cg(n_cols);
        for (int i = 0; i < n_seeds; ++i) sg[i] = ids_[seeds[i]];
        for (int i = 0; i < n_cols; ++i) cg[i] = ids_[cols[i]];
        return p_.assign_seeds(sg.data(), n_seeds, cg.data(), n_cols, kind, first_k, dist, assign);
    }
    bool triangles_batch(const int* ids, const int64_t* offsets, int n_groups, LcsBuf& out) override
    {
        std::vector<int> g((size_t)offsets[n_groups]);
        for (size_t i = 0; i < g.size(); ++i) g[i] = ids_[ids[i]];
        return p_.triangles_batch(g.data(), offsets, n_groups, out);
    }
    bool clarans(const int* ids, int n_ids, int kind, int n_medoids, int n_fixed, float fraction, int num_local,
                 int* medoids) override
    {
        std::vector<int> g(n_ids);
        for (int i = 0; i < n_ids; ++i) g[i] = ids_[ids[i]];
        return p_.clarans(g.data(), n_ids, kind, n_medoids, n_fixed, fraction, num_local, medoids);
    }

protected:
    LcsSource& p_;
    const std::vector<int>& ids_;
};

// A subset whose triangle has already been computed (as part of a batched request).
class PrecomputedSubset : public SubsetSource {
public:
    PrecomputedSubset(LcsSource& parent, const std::vector<int>& ids, std::shared_ptr<const LcsBuf> all, size_t offset)
        : SubsetSource(parent, ids), all_(std::move(all)), offset_(offset) {}
    void triangle(int r0, int r1, LcsBuf& out) override
    {
        const size_t a = (size_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2, b = (size_t)r1 * (r1 > 0 ? r1 - 1 : 0) / 2;
        out.resize(b > a ? b - a : 0, all_->wide);
        for (size_t k = a; k < b; ++k) {
            if (out.wide) out.v32[k - a] = all_->v32[offset_ + k]; else out.v16[k - a] = all_->v16[offset_ + k];
        }
    }
    const void* triangle_view() const override
    {
        return all_->wide ? (const void*)(all_->v32.data() + offset_) : (const void*)(all_->v16.data() + offset_);
    }

private:
    std::shared_ptr<const LcsBuf> all_;
    size_t offset_;
};

// Sub-trees are independent, so every split at ANY depth hands its sub-trees to one shared pool
// (largest first) and then works the pool itself until its own sub-trees are finished.  The
// reference parallelises the top-level split only, which leaves one thread to grind through a
// dominant cluster; results do not depend on who builds a sub-tree or when.
class TaskPool {
public:
    TaskPool(int n_threads, int leaf_max) : leaf_max_(leaf_max)
    {
        for (int w = 1; w < n_threads; ++w)
            workers_.emplace_back([this] {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    if (!stop_ && nothing_queued()) {
                        g_timeline.note(Timeline::IDLE, +1);
                        cv_.wait(lk, [this] { return stop_ || !nothing_queued(); });
                        g_timeline.note(Timeline::IDLE, -1);
                    }
                    if (stop_) return;
                    lk.unlock();
                    g_cpu.acquire();
                    lk.lock();
                    if (!nothing_queued()) run_top(lk);
                    g_cpu.release();
                }
            });
    }
    ~TaskPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    // A batch of tasks whose completion one caller waits for.
    struct Group {
        int remaining = 0;
        std::string error;
    };
    // `leaf_work`: the task ends in host work on data that is (or is about to be) there -- a batch of leaf matrices, leaf
    // trees.  Such tasks go before any split: a split's thread waits for the GPU most of its time and keeps no core busy,
    // so leaf work that is left to the end runs on the cores alone while the GPU idles (3 x 10^6 sequences: the last
    // 0.25 s of a 1.15 s stage, profiles/c5_timeline_r05.txt) -- but on at most `leaf_max` threads at a time while splits are
    // waiting, so that the searches of the splits keep the threads they need.
    void submit(Group& g, size_t weight, std::function<void()> fn, bool leaf_work = false)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            ++g.remaining;
            std::vector<Item>& q = leaf_work ? q_leaf_ : q_split_;
            q.push_back(Item{leaf_work, weight, seq_++, &g, std::move(fn)});
            std::push_heap(q.begin(), q.end());
        }
        cv_.notify_one();
    }
    // Work the pool (any group's tasks) until every task of `g` has finished.
    void wait(Group& g)
    {
        std::unique_lock<std::mutex> lk(mu_);
        while (g.remaining > 0) {
            if (!nothing_queued()) {
                run_top(lk);
                continue;
            }
            g_cpu.release(); // idle until a sub-task finishes or new work arrives
            g_timeline.note(Timeline::IDLE, +1);
            cv_.wait(lk, [&] { return g.remaining == 0 || !nothing_queued(); });
            g_timeline.note(Timeline::IDLE, -1);
            lk.unlock();
            g_cpu.acquire();
            lk.lock();
        }
        if (!g.error.empty()) throw std::runtime_error(g.error);
    }

private:
    struct Item {
        bool leaf_work;
        size_t weight;
        uint64_t seq;
        Group* group;
        std::function<void()> fn;
        bool operator<(const Item& o) const { return weight != o.weight ? weight < o.weight : seq > o.seq; }
    };
    bool nothing_queued() const { return q_leaf_.empty() && q_split_.empty(); }
    void run_top(std::unique_lock<std::mutex>& lk)
    { // called with the lock held; runs the heaviest queued task of the kind whose turn it is, unlocked
        const bool leaf = !q_leaf_.empty() && (leaf_running_ < leaf_max_ || q_split_.empty());
        std::vector<Item>& q = leaf ? q_leaf_ : q_split_;
        std::pop_heap(q.begin(), q.end());
        Item it = std::move(q.back());
        q.pop_back();
        if (leaf) ++leaf_running_;
        lk.unlock();
        std::string err;
        try {
            it.fn();
        } catch (const std::exception& e) {
            err = e.what();
        }
        lk.lock();
        if (leaf) --leaf_running_;
        if (!err.empty() && it.group->error.empty()) it.group->error = err;
        if (--it.group->remaining == 0) cv_.notify_all();
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Item> q_leaf_, q_split_;
    int leaf_running_ = 0;
    const int leaf_max_;
    std::vector<std::thread> workers_;
    uint64_t seq_ = 0;
    bool stop_ = false;
};

template <Distance D>
struct FastTree {
    LcsSource& src;
    GT partial;
    FastTreeParams prm;
    TaskPool* pool;
    Transform<float, D> transform; // FastTree uses float distances throughout
    struct LevelScratch { // evaluate_level_batched's arrays over all members of a level's splits, kept from level to level
        std::vector<int, NoInit<int>> sample_global, cols, assign;
        std::vector<float, NoInit<float>> dist;
    };
    LevelScratch scratch;

    // distances of subset member `ref_local` to every member: calculateDistanceVector(ref, all)
    void row_distances(const std::vector<int>& ids, int ref_local, float* out)
    {
        LcsBuf buf;
        const int ref = ids[ref_local];
        {
            Scope t(g_phase.lcs, Timeline::LCS);
            OffCpu w;
            src.rect(&ref, 1, ids.data(), (int)ids.size(), buf);
        }
        const uint32_t len_ref = src.length(ref);
        for (size_t j = 0; j < ids.size(); ++j) out[j] = transform(buf[j], len_ref, src.length(ids[j]));
    }

    int random_seeds(const std::vector<int>& ids, int n_seeds, std::vector<int>& seed_ids, float* dist_row)
    { // FastTree::randomSeeds, FastTree.cpp:335-362
        const int n = (int)ids.size();
        row_distances(ids, 0, dist_row);
        std::mt19937 mt;
        std::vector<int> rnd(n);
        std::iota(rnd.begin(), rnd.end(), 0);
        const size_t furthest = std::max_element(dist_row + 1, dist_row + n) - dist_row;
        std::swap(rnd[1], rnd[furthest]);
        partial_shuffle(rnd.data() + 2, rnd.data() + n_seeds, rnd.data() + n, mt);
        seed_ids.assign(rnd.begin(), rnd.begin() + n_seeds);
        std::sort(seed_ids.begin(), seed_ids.end());
        return n_seeds;
    }

    int cluster_seeds(const std::vector<int>& ids, int n_seeds, int n_samples, std::vector<int>& seed_ids, uint32_t seed)
    { // FastTree::clusterSeeds, FastTree.cpp:366-436.  (Its first statement, the distances of member 0 to all members, only
      // initialises the assignment sweep of makeEvaluation: make_evaluation below takes care of it.)
        const int n = (int)ids.size();
        std::vector<int> sample_ids;
        std::vector<int> sample_global;
        if (n_samples >= n) {
            n_samples = n;
            sample_global = ids;
        } else {
            std::mt19937 mt(seed);
            std::vector<int> rnd(n);
            std::iota(rnd.begin(), rnd.end(), 0);
            partial_shuffle(rnd.data() + 1, rnd.data() + n_samples, rnd.data() + n, mt);
            sample_ids.assign(rnd.begin(), rnd.begin() + n_samples);
            std::sort(sample_ids.begin(), sample_ids.end());
            sample_global.resize(n_samples);
            for (int j = 0; j < n_samples; ++j) sample_global[j] = ids[sample_ids[j]];
        }
        seed_ids.assign(n_seeds, 0);
        bool on_device;
        {   // sample matrix + CLARANS inside the engine when it offers that
            Scope t(g_phase.clarans, Timeline::CLARANS);
            OffCpu w;
            on_device = src.clarans(sample_global.data(), n_samples, (int)D, n_seeds, 1, prm.cluster_fraction,
                                    prm.cluster_iters, seed_ids.data());
        }
        if (!on_device) {
            // sample distance matrix: calculateDistanceMatrix over the samples (float)
            std::vector<float> dist((size_t)n_samples * (n_samples - 1) / 2);
            {
                SubsetSource sub(src, sample_global);
                LcsBuf buf;
                {
                    Scope t(g_phase.lcs, Timeline::LCS);
                    OffCpu w;
                    sub.triangle(0, n_samples, buf);
                }
                for (int i = 1; i < n_samples; ++i)
                    for (int j = 0; j < i; ++j)
                        dist[tri(i, j)] = transform(buf[tri(i, j)], sub.length(i), sub.length(j));
            }
            Scope t(g_phase.clarans, Timeline::CLARANS);
            Clarans{prm.cluster_fraction, prm.cluster_iters}(dist.data(), n_samples, n_seeds, 1, seed_ids.data());
        }
        if (!sample_ids.empty())
            for (int k = 0; k < n_seeds; ++k) seed_ids[k] = sample_ids[seed_ids[k]];
        return n_seeds;
    }

    float make_evaluation(const std::vector<int>& ids, int eval_num, int& n_seeds, std::vector<int>& seed_ids,
                          std::vector<int>& assignments)
    { // FastTree::makeEvaluation, FastTree.cpp:271-331
        const int n = (int)ids.size();
        std::vector<float> dist_row(n);
        const uint32_t seed = eval_num == 0 ? std::mt19937::default_seed : (uint32_t)std::hash<uint32_t>()((uint32_t)eval_num);
        // the row of member 0 (= seed 0): randomSeeds needs it to choose its seeds; after clusterSeeds it is only the
        // start of the sweep -- one request fewer per split when the engine does the sweep: seed 0 joins it, from +inf
        bool have_row0 = false;
        if (!prm.use_clustering) {
            n_seeds = random_seeds(ids, prm.subtree_size, seed_ids, dist_row.data());
            have_row0 = true;
        } else {
            n_seeds = cluster_seeds(ids, prm.subtree_size, prm.sample_size, seed_ids, seed);
            if (seed_ids[0] != 0) { // (cannot happen: slot 0 of the search is pinned to member 0 -- but then the row is needed)
                row_distances(ids, 0, dist_row.data());
                have_row0 = true;
            }
        }

        assignments.assign(n, 0);
        // seeds 1.. x all, in column chunks to bound host memory; per column the seeds are visited
        // in increasing k, exactly as the reference's row-by-row sweep
        std::vector<int> refs(n_seeds - 1);
        for (int k = 1; k < n_seeds; ++k) refs[k - 1] = ids[seed_ids[k]];
        bool on_device = false;
        if (have_row0) {
            if (n_seeds > 1) { // the sweep inside the engine when it offers that: 8 bytes per column come back
                Scope t(g_phase.assign, Timeline::ASSIGN);
                OffCpu w;
                on_device = src.assign_seeds(refs.data(), n_seeds - 1, ids.data(), n, (int)D, 1, dist_row.data(), assignments.data());
            }
        } else {
            // d(seed 0, j) < +inf for every j, so starting the sweep one seed earlier from +inf leaves exactly the row the
            // reference starts from (assignment 0), then the same strict-< updates in the same order
            std::vector<int> all_refs(n_seeds);
            for (int k = 0; k < n_seeds; ++k) all_refs[k] = ids[seed_ids[k]];
            std::fill(dist_row.begin(), dist_row.end(), std::numeric_limits<float>::infinity());
            {
                Scope t(g_phase.assign, Timeline::ASSIGN);
                OffCpu w;
                on_device = src.assign_seeds(all_refs.data(), n_seeds, ids.data(), n, (int)D, 0, dist_row.data(), assignments.data());
            }
            if (!on_device) row_distances(ids, 0, dist_row.data()); // no sweep in the engine: the reference's own first statement
        }
        const int chunk = 1 << 18;
        LcsBuf buf;
        for (int c0 = 0; c0 < n && n_seeds > 1 && !on_device; c0 += chunk) {
            const int c1 = std::min(n, c0 + chunk);
            {
                Scope t(g_phase.lcs, Timeline::LCS);
                OffCpu w;
                src.rect(refs.data(), n_seeds - 1, ids.data() + c0, c1 - c0, buf);
            }
            Scope t2(g_phase.assign, Timeline::ASSIGN);
            for (int k = 1; k < n_seeds; ++k) {
                const uint32_t len_k = src.length(refs[k - 1]);
                for (int j = c0; j < c1; ++j) {
                    const float d = transform(buf[(size_t)(k - 1) * (c1 - c0) + (j - c0)], len_k, src.length(ids[j]));
                    if (d < dist_row[j]) {
                        dist_row[j] = d;
                        assignments[j] = k;
                    }
                }
            }
        }
        return std::accumulate(dist_row.begin(), dist_row.end(), 0.0f);
    }

    bool is_leaf(size_t n) const { return !(prm.use_clustering ? (int)n > prm.threshold : (int)n > prm.subtree_size); }

    // ---- FastTree::doStep (reference tree/FastTree.cpp:56-266), LEVEL BY LEVEL -------------------------------------------
    // The reference recurses: split a subset into its seeds' clusters, build every cluster's sub-tree (recursively),
    // stitch them by a tree over the seeds.  Nothing a subset computes depends on its siblings, and the node ids of its
    // sub-tree follow from sizes alone: a subset of m members entered with `top` owns the ids [top, top + m - 1) -- its
    // clusters' sub-trees one after the other in seed order (a cluster of s members: s - 1 nodes, none for s == 1), then the
    // n_seeds - 1 nodes of the tree over the seeds.  So the recursion is walked breadth first: ALL splits of a level are
    // evaluated together -- their samples' searches share one launch wave on the device, their seed assignments one
    // batched call (LcsSource::clarans_batch / assign_seeds_batch) -- and every finished piece (a leaf's tree, a split's
    // seed tree) is written straight to its place in the final tree.  Leaf and seed trees are host work on LCS triangles
    // that arrive in batched requests: they go to the task pool as they appear and run beside the next level.
    struct Subset {
        std::vector<int> ids; // global ids
        int top;              // id of the first internal node of its sub-tree
    };
    // a tree to build with the partial generator over `ids`, written to tree[base ...]: local leaf x stands for node leaf_of[x]
    struct Piece {
        std::vector<int> ids, leaf_of;
        int base;
    };
    struct Evaluation {
        float cost = std::numeric_limits<float>::max();
        int n_seeds = -1;
        std::vector<int> seed_ids, assignments; // local to the subset
    };

    void place_piece(const Piece& pc, LcsSource& sub, tree_structure& tree)
    {
        const int m = (int)pc.ids.size();
        tree_structure local;
        {
            Scope t(g_phase.partial, Timeline::PARTIAL);
            build_tree_partial(sub, partial, D, local);
        }
        for (int node = 0; node < m - 1; ++node) {
            const node_t& nd = local[(size_t)node];
            tree[(size_t)pc.base + node] = node_t(nd.first < m ? pc.leaf_of[(size_t)nd.first] : nd.first - m + pc.base,
                                                  nd.second < m ? pc.leaf_of[(size_t)nd.second] : nd.second - m + pc.base);
        }
    }

    // the pieces' LCS triangles in batched requests (one engine call per ~24 M pairs), their trees in tasks of ~1024 members
    // (a leaf of 30 members is 20 us of work -- a task each would spend as long in the pool's queue as in the tree)
    void submit_pieces(std::vector<std::shared_ptr<Piece>>& pieces, TaskPool::Group& group, tree_structure& tree)
    {
        const size_t batch_pairs = (size_t)24 << 20;
        std::vector<std::shared_ptr<Piece>> batch;
        size_t pairs = 0, members = 0;
        auto flush = [&] {
            if (batch.empty()) return;
            pool->submit(group, members, [this, batch, &group, &tree] {
                std::vector<int> ids;
                std::vector<int64_t> offs(1, 0);
                for (const auto& pc : batch) {
                    ids.insert(ids.end(), pc->ids.begin(), pc->ids.end());
                    offs.push_back((int64_t)ids.size());
                }
                auto buf = std::make_shared<LcsBuf>(); // (kept and used again, these buffers made the stage SLOWER: 0.83 against 0.75 s, profiles/c5_stage_r06.txt)
                bool have;
                {
                    Scope tm(g_phase.lcs, Timeline::LCS);
                    OffCpu w;
                    g_leaf_requests.acquire();
                    struct Back {
                        ~Back() { g_leaf_requests.release(); }
                    } back;
                    have = src.triangles_batch(ids.data(), offs.data(), (int)batch.size(), *buf);
                }
                struct Part { std::shared_ptr<Piece> pc; size_t off; };
                std::vector<Part> parts;
                size_t off = 0, part_members = 0;
                auto submit_parts = [&] {
                    if (parts.empty()) return;
                    pool->submit(group, part_members, [this, parts, have, buf, &tree] {
                        FastTree<D> worker{src, partial, prm, pool, {}, {}};
                        for (const Part& pt : parts) {
                            if (have) {
                                PrecomputedSubset sub(src, pt.pc->ids, buf, pt.off);
                                worker.place_piece(*pt.pc, sub, tree);
                            } else {
                                SubsetSource sub(src, pt.pc->ids);
                                worker.place_piece(*pt.pc, sub, tree);
                            }
                        }
                    }, true);
                    parts.clear();
                    part_members = 0;
                };
                for (const auto& pc : batch) {
                    const size_t m = pc->ids.size();
                    parts.push_back(Part{pc, off});
                    part_members += m;
                    off += m * (m - 1) / 2;
                    if (part_members >= 1024) submit_parts();
                }
                submit_parts();
            }, true);
            batch.clear();
            pairs = members = 0;
        };
        for (auto& pc : pieces) {
            const size_t m = pc->ids.size();
            batch.push_back(pc);
            pairs += m * (m - 1) / 2;
            members += m;
            if (pairs >= batch_pairs) flush();
        }
        flush();
        pieces.clear();
    }

    // the sample of one evaluation (FastTree::clusterSeeds, FastTree.cpp:366-411): local ids, sorted; empty = every member
    void choose_sample(int n, int n_samples, uint32_t seed, std::vector<int>& sample_ids)
    {
        sample_ids.clear();
        if (n_samples >= n) return;
        std::mt19937 mt(seed);
        static thread_local std::vector<int, NoInit<int>> rnd; // (kept per thread: 12 MB of first touches for the top split otherwise)
        rnd.resize((size_t)n);
        std::iota(rnd.begin(), rnd.end(), 0);
        partial_shuffle(rnd.data() + 1, rnd.data() + n_samples, rnd.data() + n, mt);
        sample_ids.assign(rnd.begin(), rnd.begin() + n_samples);
        std::sort(sample_ids.begin(), sample_ids.end());
    }

    // FastTree::makeEvaluation for every split of a level and every evaluation number at once, through the source's batched
    // calls.  False = the source does not offer them (nothing has been computed).
    bool evaluate_level_batched(const std::vector<Subset*>& splits, std::vector<Evaluation>& best)
    {
        if (!prm.use_clustering || host_test("no_level_batch")) return false;
        const int n_splits = (int)splits.size(), n_evals = prm.num_evaluations, n_jobs = n_splits * n_evals;
        const int k = prm.subtree_size;
        auto t_mark = std::chrono::steady_clock::now();
        double t_part[5] = {0};
        auto lap = [&](int what) {
            const auto t = std::chrono::steady_clock::now();
            t_part[what] += std::chrono::duration<double>(t - t_mark).count();
            t_mark = t;
        };
        // 1. the samples
        std::vector<std::vector<int>> sample_ids((size_t)n_jobs);
        parallel_for(n_jobs, [&](int j) {
            const int s = j / n_evals, eval = j % n_evals;
            const uint32_t seed = eval == 0 ? std::mt19937::default_seed : (uint32_t)std::hash<uint32_t>()((uint32_t)eval);
            choose_sample((int)splits[(size_t)s]->ids.size(), prm.sample_size, seed, sample_ids[(size_t)j]);
        });
        std::vector<int64_t> off((size_t)n_jobs + 1, 0);
        for (int j = 0; j < n_jobs; ++j) {
            const size_t m = sample_ids[(size_t)j].empty() ? splits[(size_t)(j / n_evals)]->ids.size() : sample_ids[(size_t)j].size();
            off[(size_t)j + 1] = off[(size_t)j] + (int64_t)m;
        }
        if (host_test("no_level_scratch")) scratch = LevelScratch(); // (measurements: fresh arrays every level, as it was)
        auto& sample_global = scratch.sample_global; // (the level's large arrays are kept from level to level: no first touch, no zeros)
        sample_global.resize((size_t)off[(size_t)n_jobs]);
        parallel_for(n_jobs, [&](int j) {
            const std::vector<int>& ids = splits[(size_t)(j / n_evals)]->ids;
            int* out = sample_global.data() + off[(size_t)j];
            if (sample_ids[(size_t)j].empty()) std::copy(ids.begin(), ids.end(), out);
            else for (size_t t = 0; t < sample_ids[(size_t)j].size(); ++t) out[t] = ids[(size_t)sample_ids[(size_t)j][t]];
        });
        lap(0);
        // 2. every sample's medoids
        std::vector<int> n_medoids((size_t)n_jobs, k), medoids((size_t)n_jobs * k);
        {   // (this thread is the recursion's critical path: it keeps its core slot while it waits for the engine -- getting it
            //  back from sixteen leaf-tree tasks cost up to 70 ms a level)
            Scope t(g_phase.clarans, Timeline::CLARANS);
            if (!src.clarans_batch(sample_global.data(), off.data(), n_jobs, (int)D, n_medoids.data(), 1, prm.cluster_fraction,
                                   prm.cluster_iters, medoids.data()))
                return false;
        }
        lap(1);
        // 3. every evaluation's seed sweep: seeds x members of its split, from scratch (d(seed 0, j) < +inf for every j, so
        //    starting one seed earlier from +inf leaves the row the reference starts from, FastTree.cpp:309-324)
        std::vector<int64_t> seed_off((size_t)n_jobs + 1, 0), col_off((size_t)n_jobs + 1, 0);
        for (int j = 0; j < n_jobs; ++j) {
            seed_off[(size_t)j + 1] = seed_off[(size_t)j] + k;
            col_off[(size_t)j + 1] = col_off[(size_t)j] + (int64_t)splits[(size_t)(j / n_evals)]->ids.size();
        }
        std::vector<int> seeds_global((size_t)n_jobs * k), seeds_local((size_t)n_jobs * k);
        auto& cols = scratch.cols;
        cols.resize((size_t)col_off[(size_t)n_jobs]);
        std::atomic<bool> odd{false};
        parallel_for(n_jobs, [&](int j) {
            const std::vector<int>& ids = splits[(size_t)(j / n_evals)]->ids;
            for (int q = 0; q < k; ++q) {
                const int m = medoids[(size_t)j * k + q];
                const int local = sample_ids[(size_t)j].empty() ? m : sample_ids[(size_t)j][(size_t)m];
                seeds_local[(size_t)j * k + q] = local;
                seeds_global[(size_t)j * k + q] = ids[(size_t)local];
            }
            if (seeds_local[(size_t)j * k] != 0) odd = true; // (cannot happen: slot 0 of the search is pinned to member 0)
            std::copy(ids.begin(), ids.end(), cols.data() + col_off[(size_t)j]);
        });
        if (odd) return false; // the split-by-split form knows what the reference does then
        auto& dist = scratch.dist;
        auto& assign = scratch.assign;
        dist.resize(cols.size());
        assign.resize(cols.size());
        lap(2);
        {
            Scope t(g_phase.assign, Timeline::ASSIGN);
            if (!src.assign_seeds_batch(seeds_global.data(), seed_off.data(), cols.data(), col_off.data(), n_jobs, (int)D, dist.data(),
                                        assign.data()))
                return false;
        }
        lap(3);
        // 4. the cheapest evaluation of every split (the first one among equals: strict <, FastTree.cpp:126-138)
        std::vector<float> cost((size_t)n_jobs);
        parallel_for(n_jobs, [&](int j) {
            cost[(size_t)j] = std::accumulate(dist.begin() + col_off[(size_t)j], dist.begin() + col_off[(size_t)j + 1], 0.0f);
        });
        parallel_for(n_splits, [&](int s) {
            Evaluation& b = best[(size_t)s];
            for (int eval = 0; eval < n_evals; ++eval) {
                const int j = s * n_evals + eval;
                if (!(cost[(size_t)j] < b.cost)) continue;
                b.cost = cost[(size_t)j];
                b.n_seeds = k;
                b.seed_ids.assign(seeds_local.begin() + (size_t)j * k, seeds_local.begin() + (size_t)(j + 1) * k);
                b.assignments.assign(assign.begin() + col_off[(size_t)j], assign.begin() + col_off[(size_t)j + 1]);
            }
        });
        lap(4);
        if (profile_on())
            fprintf(stderr, "fasttree.level parts: %d evaluations: samples %.3f s, searches %.3f s, seed lists %.3f s, assignment %.3f s, costs %.3f s\n", n_jobs,
                    t_part[0], t_part[1], t_part[2], t_part[3], t_part[4]);
        return true;
    }

    // fn(i) for i in [0, count): the caller and up to n_threads - 1 helpers from the pool take the indices.  The caller never
    // picks up other tasks meanwhile (a leaf batch waits for the GPU for tens of milliseconds), and helpers that get their turn
    // late find nothing left to do.
    template <class Fn>
    void parallel_for(int count, Fn fn)
    {
        if (!pool || count < 2 || prm.n_threads < 2) {
            for (int i = 0; i < count; ++i) fn(i);
            return;
        }
        struct Shared {
            std::atomic<int> next{0};
            int done = 0, count = 0;
            std::function<void(int)> fn;
            std::mutex mu;
            std::condition_variable cv;
            std::string error;
            TaskPool::Group group;
            void work()
            {
                for (int i = next++; i < count; i = next++) {
                    std::string err;
                    try {
                        fn(i);
                    } catch (const std::exception& e) {
                        err = e.what();
                    }
                    std::lock_guard<std::mutex> lk(mu);
                    if (!err.empty() && error.empty()) error = err;
                    if (++done == count) cv.notify_all();
                }
            }
        };
        auto st = std::make_shared<Shared>();
        st->count = count;
        st->fn = fn;
        const int helpers = std::min(count, prm.n_threads) - 1;
        for (int w = 0; w < helpers; ++w) pool->submit(st->group, (size_t)1 << 40, [st] { st->work(); }); // (heaviest: next in the queue)
        st->work();
        std::unique_lock<std::mutex> lk(st->mu);
        st->cv.wait(lk, [&] { return st->done == st->count; });
        if (!st->error.empty()) throw std::runtime_error(st->error);
    }

    void run_levels(int n, tree_structure& tree)
    {
        std::vector<std::unique_ptr<Subset>> frontier;
        frontier.emplace_back(new Subset{std::vector<int>((size_t)n), n});
        std::iota(frontier[0]->ids.begin(), frontier[0]->ids.end(), 0);
        TaskPool::Group trees; // the leaf and seed trees of all levels
        struct Drain { // an error on the way out must not leave tasks behind that write to `tree` and count down `trees`
            TaskPool* pool;
            TaskPool::Group& g;
            ~Drain()
            {
                if (!pool) return;
                try { pool->wait(g); } catch (...) {}
            }
        } drain{pool, trees};
        std::vector<std::shared_ptr<Piece>> pieces;
        for (int depth = 0; !frontier.empty(); ++depth) {
            const auto t_level = std::chrono::steady_clock::now();
            std::vector<Subset*> splits;
            for (auto& sp : frontier) {
                if (is_leaf(sp->ids.size())) {
                    auto pc = std::make_shared<Piece>();
                    pc->ids = sp->ids;
                    pc->leaf_of = std::move(sp->ids);
                    pc->base = sp->top;
                    pieces.push_back(std::move(pc));
                } else
                    splits.push_back(sp.get());
            }
            if (pool && !host_test("leaves_last")) submit_pieces(pieces, trees, tree); // this level's leaves: their triangles and trees run beside what follows
            const int n_splits = (int)splits.size();
            std::vector<Evaluation> best((size_t)n_splits);
            if (!splits.empty() && !evaluate_level_batched(splits, best)) {
                parallel_for(n_splits, [&](int s) { // split by split: FastTree::makeEvaluation as it stands
                    FastTree<D> worker{src, partial, prm, pool, {}, {}};
                    Evaluation& b = best[(size_t)s];
                    for (int eval = 0; eval < prm.num_evaluations; ++eval) {
                        int ns;
                        std::vector<int> sd, as;
                        const float cost = worker.make_evaluation(splits[(size_t)s]->ids, eval, ns, sd, as);
                        if (cost < b.cost) {
                            b.cost = cost;
                            b.n_seeds = ns;
                            b.seed_ids.swap(sd);
                            b.assignments.swap(as);
                        }
                    }
                });
            }
            // the clusters of every split: the next level's subsets, and the split's own tree over its seeds
            std::vector<std::vector<std::unique_ptr<Subset>>> children((size_t)n_splits);
            std::vector<std::shared_ptr<Piece>> seed_trees((size_t)n_splits);
            parallel_for(n_splits, [&](int s) {
                Subset& sp = *splits[(size_t)s];
                Evaluation& b = best[(size_t)s];
                if (b.n_seeds < 0) throw std::runtime_error("FastTree: no evaluation produced a finite cost");
                const int n_seeds = b.n_seeds, m = (int)sp.ids.size();
                for (int q = 0; q < n_seeds; ++q) b.assignments[(size_t)b.seed_ids[(size_t)q]] = q; // seeds belong to themselves
                std::vector<int> size((size_t)n_seeds, 0);
                for (int j = 0; j < m; ++j) ++size[(size_t)b.assignments[(size_t)j]];
                std::vector<std::unique_ptr<Subset>> groups((size_t)n_seeds);
                int top = sp.top;
                auto pc = std::make_shared<Piece>();
                pc->ids.resize((size_t)n_seeds);
                pc->leaf_of.resize((size_t)n_seeds);
                for (int q = 0; q < n_seeds; ++q) {
                    const int seed = sp.ids[(size_t)b.seed_ids[(size_t)q]];
                    pc->ids[(size_t)q] = seed;
                    pc->leaf_of[(size_t)q] = seed;
                    if (size[(size_t)q] > 1) {
                        groups[(size_t)q].reset(new Subset{{}, top});
                        groups[(size_t)q]->ids.reserve((size_t)size[(size_t)q]);
                        top += size[(size_t)q] - 1;
                        pc->leaf_of[(size_t)q] = top - 1; // the cluster's root: the last node of its range
                    }
                }
                pc->base = top;
                for (int j = 0; j < m; ++j) {
                    auto& g = groups[(size_t)b.assignments[(size_t)j]];
                    if (g) g->ids.push_back(sp.ids[(size_t)j]);
                }
                for (auto& g : groups)
                    if (g) children[(size_t)s].push_back(std::move(g));
                seed_trees[(size_t)s] = std::move(pc);
                if (depth == 0 && prm.top_seeds) *prm.top_seeds = seed_trees[(size_t)s]->ids; // notifySeedsSelected(seeds, depth 0), FastTree.cpp:120-123
            });
            std::vector<std::unique_ptr<Subset>> next;
            for (int s = 0; s < n_splits; ++s) {
                for (auto& c : children[(size_t)s]) next.push_back(std::move(c));
                pieces.push_back(std::move(seed_trees[(size_t)s]));
            }
            if (profile_on())
                fprintf(stderr, "fasttree.level %d: %zu subsets, %d splits -> %zu subsets, %.3f s\n", depth, frontier.size(), n_splits, next.size(),
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_level).count());
            frontier.swap(next);
        }
        if (pool) {
            const auto t_tail = std::chrono::steady_clock::now();
            submit_pieces(pieces, trees, tree);
            pool->wait(trees);
            if (profile_on())
                fprintf(stderr, "fasttree.tail: %.3f s after the last level\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tail).count());
        } else {
            for (auto& pc : pieces) {
                SubsetSource sub(src, pc->ids);
                place_piece(*pc, sub, tree);
            }
        }
    }
};

template <Distance D>
void run_fast(LcsSource& src, GT partial, const FastTreeParams& p, tree_structure& tree)
{
    const int n = src.n();
    const auto t_in = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    if (n >= 2) tree.reserve((size_t)2 * n - 1); // (one block: assign + resize would fill n nodes, move them and fill the rest)
    tree.assign(n, node_t(-1, -1));
    if (n < 2) return;
    tree.resize((size_t)2 * n - 1, node_t(-1, -1));
    const double t_tree = since(t_in);
    // `n_threads` cores' worth of host work, two and a half times as many threads to keep GPU requests in flight
    // (trees.h, fasttree_pool_threads)
    const int n_cpu = std::max(1, p.n_threads);
    int n_pool = host_test_int("pool", fasttree_pool_threads(n_cpu)); // (FAMSA_HOST_TEST pool=N: sweeps)
    src.expect_threads(4); // (the engine's lanes: the level-by-level walk asks from this thread, the leaf batches from three others)
    const double t_lanes = since(t_in);
    g_cpu.reset(n_pool > n_cpu ? n_cpu + 1 : 0); // (+ 1: this thread's own, kept through the levels)
    g_cpu.acquire();
    struct Giveback {
        ~Giveback() { g_cpu.reset(0); }
    } giveback;
    double t_setup = 0, t_levels = 0;
    const auto t_out = [&] {
        TaskPool pool(n_pool, host_test_int("leafmax", fasttree_leaf_threads(n_pool))); // (FAMSA_HOST_TEST leafmax=N: sweeps)
        FastTree<D> ft{src, partial, p, &pool, {}, {}};
        t_setup = since(t_in);
        const auto t0 = std::chrono::steady_clock::now();
        ft.run_levels(n, tree);
        t_levels = since(t0);
        std::thread([held = std::move(ft.scratch)]() mutable { (void)held; }).detach(); // (its 50 MB are not unmapped by this thread either)
        return std::chrono::steady_clock::now();
    }();
    if (profile_on())
        fprintf(stderr, "fasttree.stage: before the levels %.3f s (the empty tree %.3f, the engine's lanes %.3f, the pool %.3f), levels %.3f s, pool shut down %.3f s\n",
                t_setup, t_tree, t_lanes - t_tree, t_setup - t_lanes, t_levels, since(t_out));
}

} // namespace

void clarans_host(const float* distances, int n_elems, int n_medoids, int n_fixed, float explore_fraction, int num_local,
                  int* medoids)
{
    Clarans{explore_fraction, num_local}(distances, n_elems, n_medoids, n_fixed, medoids);
}

void build_tree_fast(LcsSource& src, GT partial, Distance dist, const FastTreeParams& p, tree_structure& tree)
{
    if (partial == GT::chained) throw std::runtime_error("Error: Illegal guide tree method."); // msa.cpp:170: no generator to wrap
    if (partial == GT::MST_Prim) partial = GT::SLINK; // reference msa.cpp:134: MST+Prim is not a partial generator
    if (dist == Distance::indel_div_lcs) run_fast<Distance::indel_div_lcs>(src, partial, p, tree);
    else if (dist == Distance::indel075_div_lcs) run_fast<Distance::indel075_div_lcs>(src, partial, p, tree);
    else throw std::runtime_error("Error: Illegal pairwise distance measure.");
    if (profile_on()) {
        fprintf(stderr, "fasttree.lcs_calls=%.3f\nfasttree.clarans=%.3f\nfasttree.partial_trees=%.3f\nfasttree.assign=%.3f\n", g_phase.lcs, g_phase.clarans,
                g_phase.partial, g_phase.assign);
        g_phase = PhaseTimers(); // (a library caller's next tree starts from zero)
        g_timeline.dump();
    }
}

} // namespace famsa_host
