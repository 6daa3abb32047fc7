// clarans_prune_stats.cpp -- measurement aid (CPU): how many slots of a CLARANS step can have a negative delta at all?
//
// A step's delta for slot s is the sequential float sum of (a) the addends y_j < 0 of the members that are closer to the
// candidate than to their own medoid -- the same value for every slot -- and (b) the addends x_i >= 0 of the slot's other
// members (reference tree/Clustering.cpp:93-118).  With Y = sum of (a) and X_s = sum of (b), a slot with
// X_s > 1.002 |Y| has a positive delta whatever the order of the additions (recursive-summation error bound, 2048 terms),
// so only the other slots -- the set P -- need the exact walk, and a step with an empty P cannot be accepted.
// This program runs the search (the serial form of famsa_amd/host/fasttree.cpp, HostClarans) on a float distance triangle
// read from a file and prints the distribution of |P| and of the entries a filtered walk touches.
//   g++ -O2 -o /tmp/prune/stats scripts/clarans_prune_stats.cpp && /tmp/prune/stats tri.f32 n k
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <random>
#include <vector>
static inline size_t tri(size_t i, size_t j) { return i >= j ? j + i * (i - 1) / 2 : i + j * (j - 1) / 2; }
template <class UInt, class Gen> static long long det_uniform(Gen& g, int lo, int hi)
{
    const UInt diff = (UInt)hi - (UInt)lo + 1;
    const UInt bad = std::numeric_limits<UInt>::max() / diff;
    for (;;) { const UInt r = (UInt)g(); if (r / diff < bad) return (long long)((r % diff) + (UInt)lo); }
}
struct Two { float dn, ds; int an, as; };
int main(int argc, char** argv)
{
    if (argc < 4) return 1;
    const int n = atoi(argv[2]), k = atoi(argv[3]), fixed = 1;
    std::vector<float> D((size_t)n * (n - 1) / 2);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(D.data(), 4, D.size(), f) != D.size()) return 2;
    fclose(f);
    const int n_swaps = (n - k) * k, max_neighbor = n_swaps < 250 ? n_swaps : std::max((int)(0.1f * n_swaps), 250), corrected = max_neighbor / k;
    std::vector<int> member(n);
    std::iota(member.begin(), member.end(), 0);
    std::mt19937 shuffle_gen, position_gen;
    std::vector<Two> st(n);
    std::vector<float> to_slot((size_t)k * n), delta(k), X(k);
    auto two_nearest = [&](int pos, int swap_slot, float d_swap) {
        Two t{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), -1, -1};
        for (int s = 0; s < k; ++s) {
            const float d = s == swap_slot ? d_swap : to_slot[(size_t)s * n + pos];
            if (d < t.dn) { t.ds = t.dn; t.as = t.an; t.dn = d; t.an = s; } else if (d < t.ds) { t.ds = d; t.as = s; }
        }
        return t;
    };
    long steps = 0, accepts = 0, empty_b = 0, empty_P = 0, wrong = 0;
    long hist_P[8] = {0}; // |P| = 0, 1, 2, 3-4, 5-8, 9-16, 17-32, more
    double walked_filtered = 0, walked_all = 0, common = 0;
    long accepted_with_P[8] = {0};
    for (int iter = 0; iter < 2; ++iter) {
        { const long cnt = n - fixed, N = cnt - 1; int* first = member.data() + fixed;
          for (long i = 0; i < cnt; ++i) std::swap(first[i], first[det_uniform<unsigned long>(shuffle_gen, (int)i, (int)N)]); }
        for (int pos = k; pos < n; ++pos) {
            for (int s = 0; s < k; ++s) to_slot[(size_t)s * n + pos] = D[tri(member[s], member[pos])];
            st[pos] = two_nearest(pos, -1, 0.0f);
        }
        int allowed = corrected;
        for (int quiet = 0; quiet < allowed;) {
            ++quiet; ++steps;
            const int xx = (int)det_uniform<unsigned int>(position_gen, k, n - 1), x = member[xx];
            std::fill(delta.begin(), delta.end(), 0.0f);
            std::fill(X.begin(), X.end(), 0.0f);
            float Y = 0; int nb = 0;
            for (int pos = k; pos < n; ++pos) {
                if (pos == xx) continue;
                const float d = D[tri(x, member[pos])];
                const Two& t = st[pos];
                const float own = std::min(d, t.ds) - t.dn, other = d - t.dn;
                if (other < 0.0f) { for (int s = 0; s < k; ++s) delta[s] += s == t.an ? own : other; Y += other; ++nb; }
                else { delta[t.an] += own; X[t.an] += own; }
            }
            const int slot = (int)(std::min_element(delta.begin() + fixed, delta.end()) - delta.begin());
            int nP = 0; double wf = nb;
            std::vector<char> inP(k, 0);
            if (nb > 0) for (int s = fixed; s < k; ++s) if (!(X[s] > 1.002f * -Y)) { inP[s] = 1; ++nP; }
            if (nP) for (int pos = k; pos < n; ++pos) if (pos != xx && inP[st[pos].an] && !(D[tri(x, member[pos])] - st[pos].dn < 0.0f)) wf += 1;
            const int bin = nP == 0 ? 0 : nP == 1 ? 1 : nP == 2 ? 2 : nP <= 4 ? 3 : nP <= 8 ? 4 : nP <= 16 ? 5 : nP <= 32 ? 6 : 7;
            hist_P[bin]++; if (nb == 0) ++empty_b; if (nP == 0) ++empty_P;
            walked_filtered += nP ? wf : 0; walked_all += n - k - 1; common += nb;
            if (delta[slot] < 0.0f) {
                if (!inP[slot]) ++wrong; // must never happen
                accepted_with_P[bin]++;
                ++accepts;
                const int old_medoid = member[slot];
                member[slot] = x; member[xx] = old_medoid;
                for (int pos = k; pos < n; ++pos) {
                    if (pos == xx) { for (int s2 = 0; s2 < k; ++s2) to_slot[(size_t)s2 * n + pos] = D[tri(member[s2], old_medoid)]; st[pos] = two_nearest(pos, -1, 0.0f); continue; }
                    const float d = D[tri(x, member[pos])];
                    Two& t = st[pos];
                    if (t.an == slot) { if (d < t.ds) t.dn = d; else t = two_nearest(pos, slot, d); }
                    else if (d < t.dn) t = Two{d, t.dn, slot, t.an};
                    else if (t.as != slot && d < t.ds) { t.ds = d; t.as = slot; }
                    else t = two_nearest(pos, slot, d);
                    to_slot[(size_t)slot * n + pos] = d;
                }
                quiet = 0; allowed = corrected - 1;
            }
        }
    }
    printf("n=%d k=%d corrected=%d: steps=%ld accepts=%ld  (b)-entries per step %.2f  steps with no (b) entry %.1f%%  steps with empty P %.1f%%\n", n, k, corrected,
           steps, accepts, common / steps, 100.0 * empty_b / steps, 100.0 * empty_P / steps);
    static const char* names[8] = {"0", "1", "2", "3-4", "5-8", "9-16", "17-32", ">32"};
    for (int b = 0; b < 8; ++b) printf("  |P| %-6s %7.2f%% of the steps, %ld of the accepts\n", names[b], 100.0 * hist_P[b] / steps, accepted_with_P[b]);
    printf("  entries walked: all %.0f per step, filtered %.1f per step (%.1f%%); accepted slot outside P: %ld (must be 0)\n", walked_all / steps,
           walked_filtered / steps, 100.0 * walked_filtered / walked_all, wrong);
    return 0;
}
