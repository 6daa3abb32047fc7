// trees.h -- guide-tree builders over GPU-computed LCS lengths.
// Each builder restates one reference generator's algorithm on top of an LcsSource; the
// distances are produced with the reference's Transform types and operand order so that tie
// breaking and float/double rounding, hence the tree topology, are identical.
// Tree representation = the reference's tree_structure (tree/TreeDefs.h:15-16): node i < n is a
// leaf (-1,-1); internal nodes are appended as (left, right) child ids.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "lcs_source.h"
#include "transform.h"

namespace famsa_host {

using node_t = std::pair<int, int>;
using tree_structure = std::vector<node_t>;

enum class GT { MST_Prim, SLINK, UPGMA, UPGMA_modified, NJ, chained };
GT gt_from_string(const std::string& name); // "sl" | "slink" | "upgma" | "upgma_modified" | "nj" | "chained"

// Build the guide tree over src's sequences (ids 0..n-1 = the sorted unique working order).
// Result has 2n-1 nodes.  Restates (reference, src/tree/): MSTPrim.cpp:280-549 + 784-833,
// SingleLinkage.cpp:31-189, UPGMA.cpp:39-51 + 114-295, NeighborJoining.cpp:10-118.
void build_tree(LcsSource& src, GT method, Distance dist, tree_structure& tree, int n_threads);

// `-gt chained [seed]` (reference tree/TreeDefs.h:59,98; tree/Chained.h:8-35, developer builds only): a caterpillar over a
// random order of the n leaves -- node n = (idx[0], idx[1]), node n + i - 1 = (idx[i], the previous node).  No distances
// are computed.  The reference seeds its shuffle from std::random_device (a different tree on every run) and release
// builds answer "Illegal guide tree method"; here the order is a Fisher-Yates shuffle driven by mt19937(seed) through the
// reference's own det_uniform_int_distribution, so that a seed names one tree on every platform.
void build_tree_chained(int n, uint32_t seed, tree_structure& tree);

// IPartialGenerator::runPartial (reference tree/IPartialGenerator.h:13): APPEND the n-1 internal
// nodes of the tree over src's n sequences to `tree`, with local ids (leaves 0..n-1, internal
// nodes n..2n-2).  Only SLINK, UPGMA(_modified) and NJ are partial generators.
void build_tree_partial(LcsSource& src, GT method, Distance dist, tree_structure& tree);

// MedoidTree / PartTree heuristic (reference tree/FastTree.cpp + tree/Clustering.cpp): recursive
// seed selection (CLARANS k-medoids on a sample, or random seeds), assignment of every sequence to
// its nearest seed, sub-trees per cluster with `partial`, stitched by a tree over the seeds.
struct FastTreeParams { // CParams::medoid, reference core/params.h:88-97
    bool use_clustering = true; // true: -medoidtree (CLARANS), false: -parttree (random seeds)
    int subtree_size = 100;
    int sample_size = 2000;
    int num_evaluations = 1;
    int threshold = 2000;
    float cluster_fraction = 0.1f;
    int cluster_iters = 2;
    int n_threads = 1; // host cores the recursion may keep busy
    // -dump_seeds (reference msa.cpp:184-199, tree/FastTree.cpp:120-123): when set, receives the seeds the top-level
    // split ends up with (depth 0 only), as ids of the source, in seed order
    std::vector<int>* top_seeds = nullptr;
};
// threads of the recursion's task pool for `n_cpu` cores: the ones waiting for the GPU cost no core.  A quarter of them at
// most work on leaves while splits are waiting (fasttree.cpp, TaskPool).  3 x 10^6 sequences, 16 cores, tree stage: 32
// threads, leaves on any number of them 1.09-1.13 s; 32 / 8: 0.99-1.01 s; 40 / 10: 0.92-0.95 s; 48 / 16: 1.05-1.09 s
// (profiles/c5_pool_r05.txt).
inline int fasttree_pool_threads(int n_cpu) { return n_cpu > 1 ? (5 * n_cpu + 1) / 2 : 1; }
inline int fasttree_leaf_threads(int n_pool) { return n_pool > 4 ? n_pool / 4 : n_pool; }
void build_tree_fast(LcsSource& src, GT partial, Distance dist, const FastTreeParams& p, tree_structure& tree);
// the host form of the CLARANS search (used when the LcsSource does not run it itself)
void clarans_host(const float* distances, int n_elems, int n_medoids, int n_fixed, float explore_fraction, int num_local,
                  int* medoids);

// GuideTree::fromUnique (reference tree/GuideTree.cpp:146-208): re-attach removed duplicates.
void tree_from_unique(tree_structure& tree, const std::vector<int>& sorted2unique);

// NewickParser::store (reference tree/NewickParser.cpp:103-165); names[i] = id of leaf i
// (a leading '>' is dropped), every branch ":1.0", no trailing newline.
std::string tree_to_newick(const tree_structure& tree, const std::vector<std::string>& names);
std::string tree_to_newick(const tree_structure& tree, const std::vector<const char*>& names); // the same over NUL-terminated names the caller keeps

// -dist_export writer (reference tree/DistanceCalculator.cpp:11-122 with
// utils/conversion.h:109-119): rows in input order, ref = row, partner = column.
void write_distance_csv(LcsSource& src, const std::vector<std::string>& ids, Distance dist, bool square,
                        bool pid, const std::string& path);
// the number format alone (exposed for tests): returns characters written
int format_distance(double val, char* out);

} // namespace famsa_host
