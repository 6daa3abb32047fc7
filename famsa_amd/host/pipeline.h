// pipeline.h -- the slice of CFAMSA::ComputeMSA (reference msa.cpp:470-584) that surrounds the
// guide-tree generators: working order, duplicate removal, generator call, fromUnique, Newick;
// and the -dist_export early branch (msa.cpp:518-526: input order, no sort, no dedup).
#pragma once
#include <future>
#include <memory>
#include <string>

#include "lcs_source.h"
#include "seqset.h"
#include "trees.h"

namespace famsa_host {

struct Timings {
    double load_s = 0, sort_s = 0, init_s = 0, upload_s = 0, tree_s = 0, newick_s = 0, store_s = 0, kernel_ms = 0;
    long rss_load_kb = 0, rss_upload_kb = 0, rss_tree_kb = 0, rss_newick_kb = 0; // resident host memory after the stage (/proc/self/status)
    int n_records = 0, n_duplicates = 0; // input.n_sequences / input.n_duplicates of the reference's statistics (famsa.cpp:114, msa.cpp:662)
    std::string transport; // several GPUs: lcsgpu_multi_transport's report (how the contexts reached each other, which key exchange ran)
};

// Newick for the whole input (duplicates re-attached), LCS values from `src_of_unique`, whose
// sequence ids are the sorted unique working order.
// Tree-stage options of the reference CLI (core/params.h:81-104)
struct TreeOptions {
    GT method = GT::MST_Prim;
    Distance dist = Distance::indel075_div_lcs;
    bool keep_duplicates = false;
    int heuristic = 0; // 0 none, 1 -parttree, 2 -medoidtree
    FastTreeParams fast;
    uint32_t chained_seed = 0;        // -gt chained [seed]
    std::string dump_seeds_path;      // -dump_seeds <file>: ids of the top-level split's seeds, one per line (msa.cpp:184-199)
};

long resident_kb(); // VmRSS of this process
// spare: storage the caller has set up for the tree meanwhile (its content does not matter, its capacity and touched pages do)
std::string guide_tree_newick(const SeqSet& s, const WorkSet& w, LcsSource& src_of_unique, const TreeOptions& opt,
                              Timings* t = nullptr, tree_structure* spare = nullptr);

std::string guide_tree_newick_from_matrix(const SeqSet& s, const uint32_t* square_input_order, const TreeOptions& opt);
// The engine context takes ~0.2 s to create (HIP initialisation): a caller may start that early, on another
// thread, and hand the future in; otherwise it is started here, next to the sort.
using EngineFuture = std::future<std::unique_ptr<GpuLcsSource>>;
// A command-line tool that ends right after the call may skip the engine's teardown (~tens of ms of stream /
// buffer destruction that the operating system does anyway): set before calling the *_gpu functions.
extern bool g_abandon_engine_at_return;
EngineFuture start_engine(int device);
EngineFuture start_engine(const std::vector<int>& devices, int expect_threads = 0); // one context per entry (entries may repeat)
std::string guide_tree_newick_gpu(const SeqSet& s, int device, const TreeOptions& opt, Timings* t,
                                  EngineFuture* engine = nullptr);
// The same for a caller that does not need the residues afterwards: once the tree is built, `s.codes` is given back to
// the system by a background thread while the Newick is made (at 3 x 10^6 records the process otherwise carries 0.8 GB
// to its exit, and the exit takes that much longer).  Ids, offsets and lengths stay.
std::string guide_tree_newick_gpu_consuming(SeqSet& s, int device, const TreeOptions& opt, Timings* t,
                                            EngineFuture* engine = nullptr);
void dist_export_gpu(const SeqSet& s, int device, Distance dist, bool square, bool pid, const std::string& path,
                     Timings* t, EngineFuture* engine = nullptr);

} // namespace famsa_host
