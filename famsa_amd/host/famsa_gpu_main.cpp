// famsa-gpu -- the guide-tree / distance-export front end of FAMSA on the MI355X LCS engine.
// Accepts the reference CLI's flags for this path (reference core/params.cpp:136-294, help text
// :60-134):   famsa-gpu [options] <input.fasta> <output>
//   -gt <sl|slink|upgma|upgma_modified|nj|chained [seed]>   guide tree method (default sl)
//   -gt_export                               write the guide tree in Newick format and stop
//   -dist_export [-pid] [-square_matrix]     write the distance (or identity) matrix as CSV
//   -dist <indel_div_lcs|indel075_div_lcs>   distance measure (default indel075_div_lcs)
//   -keep-duplicates | -keep_duplicates, -t <n>, -v / -vv, -stats <file>, -gpu <id[,id...]> | -gpus <n>   (several GPUs: the
//                                              pair space is tiled by row blocks, FastTree batches are split between the devices)
//   -medoidtree | -parttree [-medoid_threshold n -subtree_size n -sample_size n -num_evals n -cluster_fraction f
//                            -cluster_iters n -dump_seeds <file>]
//   -shuffle <n>                             accepted and ignored (parsed by developer builds of the reference only,
//                                            core/params.cpp:242-244, and consumed nowhere)
// Everything downstream of the guide tree (profile alignment, refinement, gz I/O)
// is outside this engine's scope and is refused with an error, not silently ignored.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "pipeline.h"

using namespace famsa_host;

namespace {

bool find_switch(std::vector<std::string>& p, const std::string& name)
{
    auto it = std::find(p.begin(), p.end(), name);
    if (it == p.end()) return false;
    p.erase(it);
    return true;
}

bool find_option(std::vector<std::string>& p, const std::string& name, std::string& value)
{
    auto it = std::find(p.begin(), p.end(), name);
    if (it == p.end() || it + 1 == p.end()) return false;
    value = *(it + 1);
    p.erase(it, it + 2);
    return true;
}

void usage()
{
    std::cerr << "Usage: famsa-gpu [options] <input_file> <output_file>\n"
                 "  -gt <sl | slink | upgma | upgma_modified | nj | chained [seed]>  guide tree method (default: sl)\n"
                 "  -gt_export            export the guide tree to the output file in Newick format\n"
                 "  -dist_export          export the distance matrix to the output file in CSV format\n"
                 "  -square_matrix        generate a square distance matrix instead of the lower triangle\n"
                 "  -pid                  export pairwise identity (the number of matching residues divided\n"
                 "                        by the shorter sequence length) instead of distance\n"
                 "  -dist <measure>       indel_div_lcs | indel075_div_lcs (default)\n"
                 "  -keep-duplicates      keep duplicated sequences during tree construction (also: -keep_duplicates)\n"
                 "  -medoidtree | -parttree   MedoidTree / PartTree heuristic (with -medoid_threshold, -subtree_size,\n"
                 "                        -sample_size, -num_evals, -cluster_fraction, -cluster_iters as in FAMSA)\n"
                 "  -dump_seeds <file>    with a heuristic: the ids of the top-level split's seeds, one per line\n"
                 "  -stats <file>         the run's statistics as \"[stats]\" + key=value lines (what -v prints)\n"
                 "  -gpu <id[,id...]>     HIP device(s) (default 0); -gpus <n> = devices 0..n-1.  With several devices\n"
                 "                        the all-pairs work is tiled by row blocks over them\n"
                 "  -t <n>, -v, -vv       accepted for compatibility / verbosity\n";
}

} // namespace

int main(int argc, char** argv)
{
    try {
        std::vector<std::string> params(argv + 1, argv + argc);
        if (params.size() < 2 || find_switch(params, "-help")) {
            usage();
            return 0;
        }
        std::string aux;
        TreeOptions opt;
        std::vector<int> devices{0};
        {   // -gt <method>; "chained" may be followed by its seed (core/params.cpp:178-189)
            auto it = std::find(params.begin(), params.end(), std::string("-gt"));
            if (it != params.end() && it + 1 != params.end()) {
                aux = *(it + 1);
                it = params.erase(it, it + 2);
                if (aux == "import") throw std::runtime_error("-gt import needs the alignment stage, which is outside this tool");
                opt.method = gt_from_string(aux);
                if (opt.method == GT::chained && it != params.end() && !it->empty() &&
                    it->find_first_not_of("0123456789") == std::string::npos) {
                    opt.chained_seed = (uint32_t)std::stoul(*it);
                    params.erase(it);
                }
            }
        }
        if (find_option(params, "-dist", aux)) {
            if (aux == "indel_div_lcs") opt.dist = Distance::indel_div_lcs;
            else if (aux == "indel075_div_lcs") opt.dist = Distance::indel075_div_lcs;
            else throw std::runtime_error("Error: Illegal pairwise distance measure.");
        }
        // MedoidTree / PartTree heuristic and its parameters (reference core/params.cpp:195-208)
        if (find_switch(params, "-parttree")) opt.heuristic = 1;
        if (find_switch(params, "-medoidtree")) opt.heuristic = 2;
        if (find_option(params, "-medoid_threshold", aux)) opt.fast.threshold = std::stoi(aux);
        if (find_option(params, "-subtree_size", aux)) opt.fast.subtree_size = std::stoi(aux);
        if (find_option(params, "-sample_size", aux)) opt.fast.sample_size = std::stoi(aux);
        if (find_option(params, "-num_evals", aux)) opt.fast.num_evaluations = std::stoi(aux);
        if (find_option(params, "-dump_seeds", aux)) opt.dump_seeds_path = aux;
        std::string stats_path;
        (void)find_option(params, "-stats", stats_path);
        (void)find_option(params, "-shuffle", aux); // developer builds of the reference parse it; nothing consumes it
        if (find_option(params, "-cluster_fraction", aux)) opt.fast.cluster_fraction = std::stof(aux);
        if (find_option(params, "-cluster_iters", aux)) opt.fast.cluster_iters = std::stoi(aux);
        if (find_option(params, "-gpu", aux)) {
            devices.clear();
            for (size_t at = 0; at <= aux.size();) {
                const size_t comma = std::min(aux.find(',', at), aux.size());
                devices.push_back(std::stoi(aux.substr(at, comma - at)));
                at = comma + 1;
            }
        }
        if (find_option(params, "-gpus", aux)) {
            devices.clear();
            for (int d = 0; d < std::stoi(aux); ++d) devices.push_back(d);
            if (devices.empty()) throw std::runtime_error("-gpus needs a positive count");
        }
        // -t <n>: host worker threads (0 = half of the hardware threads, reference core/params.cpp:285-291)
        int n_threads = 0;
        if (find_option(params, "-t", aux)) n_threads = std::stoi(aux);
        if (n_threads <= 0) n_threads = default_host_threads(); // respects a container's cpu.max
        opt.fast.n_threads = n_threads;
        const bool very_verbose = find_switch(params, "-vv");
        if (very_verbose) setenv("LCSGPU_MST_COUNTS", "1", 0); // the engine's "tiles computed / let go" line (the reference's -vv prints its counts)
        const bool verbose = find_switch(params, "-v") || very_verbose;
        const bool export_tree = find_switch(params, "-gt_export");
        const bool export_dist = find_switch(params, "-dist_export");
        const bool square = find_switch(params, "-square_matrix");
        const bool pid = find_switch(params, "-pid");
        opt.keep_duplicates = find_switch(params, "-keep-duplicates");
        if (find_switch(params, "-keep_duplicates")) opt.keep_duplicates = true; // core/params.cpp:240: both spellings
        for (const char* unsupported : {"-gz", "-refine_mode", "-trim_columns", "-go", "-ge", "-r"})
            if (std::find(params.begin(), params.end(), unsupported) != params.end())
                throw std::runtime_error(std::string(unsupported) + " is outside the scope of famsa-gpu (guide tree / distance stage only)");
        if (params.size() != 2) {
            usage();
            return 1;
        }
        if (!export_tree && !export_dist)
            throw std::runtime_error("famsa-gpu covers -gt_export and -dist_export; the alignment stage is not part of this engine");

        const std::string input = params[0], output = params[1];
        const auto clock0 = std::chrono::steady_clock::now();
        const auto clock_main = clock0;
        auto since = [](std::chrono::steady_clock::time_point a) {
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        };
        // the engine wants one hardware queue per lane (lcsgpu_create sets this too, but the environment must not
        // be modified once other threads run)
        // -- capped at 16 like the library's own setting: more hardware queues cost more than they give
        {
            const int lanes = getenv("LCSGPU_LANES") ? atoi(getenv("LCSGPU_LANES")) : 16;
            setenv("GPU_MAX_HW_QUEUES", std::to_string(std::max(1, std::min(lanes, 16))).c_str(), 0);
        }
        g_abandon_engine_at_return = getenv("FAMSA_GPU_CLEAN_EXIT") == nullptr;
        // HIP initialisation runs while the input is read and sorted; the heuristics' worker threads get their lanes now
        EngineFuture engine = start_engine(devices, opt.heuristic != 0 ? fasttree_pool_threads(n_threads) : 0);
        const int device = devices[0];
        SeqSet s = load_fasta(input, n_threads);
        if (s.size() == 0) throw std::runtime_error("no sequences in " + input);
        Timings t;
        t.load_s = since(clock0);
        t.rss_load_kb = resident_kb();
        if (export_dist) {
            dist_export_gpu(s, device, opt.dist, square, pid, output, &t, &engine);
        } else {
            const std::string nwk = guide_tree_newick_gpu_consuming(s, device, opt, &t, &engine); // the residues are not needed again
            const auto clock1 = std::chrono::steady_clock::now();
            std::ofstream f(output, std::ios::binary);
            if (!f.good()) throw std::runtime_error("cannot open " + output);
            f << nwk;
            f.close();
            if (f.fail()) throw std::runtime_error("writing " + output + " failed (disk full?)"); // before the fast exit below reports success
            t.store_s = since(clock1);
        }
        if (!stats_path.empty()) {
            // the reference's statistics file (famsa.cpp:137-150, utils/statistics.h:77-87): "[stats]", then key=value lines
            // in key order.  Its keys where this tool has the figure (input.*, time.sort, time.tree_build, time.tree_store,
            // time.save, time.total), plus the engine's own.
            std::map<std::string, std::string> kv;
            auto put = [&kv](const std::string& k, double v) { kv[k] = std::to_string(v); };
            kv["input.n_sequences"] = std::to_string(s.size());
            if (!export_dist) kv["input.n_duplicates"] = std::to_string(t.n_duplicates);
            put("time.load", t.load_s);
            put("time.sort", t.sort_s);
            put("time.tree_build", t.tree_s);
            put("time.tree_store", t.newick_s + t.store_s);
            put("time.save", 0.0); // no alignment is written
            put("time.total", since(clock_main));
            put("time.gpu_init", t.init_s);
            put("time.gpu_upload", t.upload_s);
            put("gpu.lcs_kernel_ms", t.kernel_ms);
            kv["mem.after_tree_kB"] = std::to_string(t.rss_tree_kb);
            std::ofstream f(stats_path);
            f << "[stats]\n";
            for (const auto& e : kv) f << e.first << "=" << e.second << "\n";
            f.close();
            if (f.fail()) throw std::runtime_error("writing " + stats_path + " failed");
        }
        if (verbose) {
            std::cerr << "time.load=" << t.load_s << "\n"
                      << "time.sort=" << t.sort_s << "\n"
                      << "time.gpu_init=" << t.init_s << "\n"
                      << "time.gpu_upload=" << t.upload_s << "\n"
                      << "time.tree_build=" << t.tree_s << "\n"
                      << "time.newick=" << t.newick_s << "\n"
                      << "time.store=" << t.store_s << "\n"
                      << "gpu.lcs_kernel_ms=" << t.kernel_ms << "\n"
                      << "time.main_until_exit=" << since(clock_main) << "\n";
            std::cerr << "mem.after_load_kB=" << t.rss_load_kb << "\nmem.after_upload_kB=" << t.rss_upload_kb << "\nmem.after_tree_kB=" << t.rss_tree_kb
                      << "\nmem.after_newick_kB=" << t.rss_newick_kb << "\n";
            { // the host memory this process peaked at and holds now (what the system takes back after the exit)
                std::ifstream st("/proc/self/status");
                std::string line;
                while (std::getline(st, line))
                    if (line.compare(0, 6, "VmHWM:") == 0 || line.compare(0, 6, "VmRSS:") == 0)
                        std::cerr << "mem." << line.substr(0, 5) << "_kB=" << atol(line.c_str() + 6) << "\n";
            } // what the caller's wall clock adds: loading the HIP libraries before main, the teardown of the process after it
            if (!t.transport.empty()) { // several GPUs: one "gpu.transport=" line per fact
                size_t at = 0;
                while (at < t.transport.size()) {
                    const size_t nl = t.transport.find('\n', at);
                    std::cerr << "gpu.transport=" << t.transport.substr(at, nl == std::string::npos ? std::string::npos : nl - at) << "\n";
                    if (nl == std::string::npos) break;
                    at = nl + 1;
                }
            }
        }
        // The result is on disk.  Tearing the HIP runtime down (contexts, queues, code objects) costs ~0.1 s,
        // a third of a small run's wall time; the process ends here instead (FAMSA_GPU_CLEAN_EXIT=1 for
        // leak checkers).
        if (!getenv("FAMSA_GPU_CLEAN_EXIT")) {
            std::cerr.flush();
            std::cout.flush();
            fflush(nullptr);
            _exit(0);
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "[ERROR] " << e.what() << std::endl;
        return -1;
    }
}
