// msa_harness.cpp -- the reference's whole CFAMSA::ComputeMSA, timed stage by stage.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as ref_harness.cpp): built into oracle/_ref/libfamsa_msa.so from the
// reference's own sources where they lie under /root/reference (msa.cpp, msa_refinement.cpp, core/profile*.cpp,
// core/params.cpp beside the hot-path units; the vendored libs/refresh headers with the reference's own REFRESH_USE_ZLIB
// switch and the system zlib) plus this driver.  Nothing of the alignment stage is restated, replaced or shipped: this is how
// BASELINE.json's second metric clause -- "end-to-end MSA wall time vs CPU" -- is measured (scripts/e2e_msa.py):
//   CPU:  ComputeMSA as it stands (sort, guide tree with the reference's CLCSBP, progressive alignment, refinement);
//   GPU:  the SAME ComputeMSA with `-gt import <newick>` -- the tree famsa-gpu wrote for the same input -- so that the tree
//         stage's cost is famsa-gpu's and everything downstream is the reference's object code on an identical tree (the
//         alignments of both runs are compared byte for byte).
// The stage times are the reference's own timers (msa.cpp:530-619: CFAMSA::timers, read through a derived class).
// core/io_service.cpp (needs the absent libdeflate) is not part of the build: sequences are read here with the line handling
// of IOService::loadFasta (io_service.h:84-127) and the alignment is written as plain FASTA, one line per sequence.
#include "msa.h"

#include <chrono>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace {

struct Timed : CFAMSA {
    explicit Timed(CParams& p) : CFAMSA(p) {}
    double t(int which) { return timers[which].GetElapsedTime(); }
    double sort() { return t(TIMER_SORTING); }
    double tree() { return t(TIMER_TREE_BUILD); }
    double align() { return t(TIMER_ALIGNMENT); }
    double refine() { return t(TIMER_REFINMENT); }
};

void read_fasta(const std::string& path, std::vector<CSequence>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("cannot open " + path);
    std::string s, id, seq;
    int no = 0;
    auto flush = [&] {
        if (!id.empty() && !seq.empty()) out.emplace_back(id, seq, no++, nullptr);
        seq.clear();
    };
    while (std::getline(f, s)) {
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        if (s.empty()) continue;
        if (s[0] == '>') {
            flush();
            id = s;
        } else
            seq += s;
    }
    flush();
}

} // namespace

extern "C" {

// argv-style options as the famsa command line takes them (e.g. "-gt upgma -t 8", or "-gt import tree.dnd -t 8"), without the
// input / output file names.  times[0..5] = load, sort, tree, alignment, refinement, whole ComputeMSA (seconds).  The alignment
// (input order) goes to `alignment_path` if not NULL.  Returns the number of aligned sequences, negative on error.
int msa_run(const char* fasta, const char* options, double* times, const char* alignment_path, char* error, int error_cap)
{
    try {
        std::vector<std::string> words{"famsa"};
        {
            std::string w;
            for (const char* p = options ? options : ""; ; ++p) {
                if (*p == ' ' || *p == 0) {
                    if (!w.empty()) words.push_back(w);
                    w.clear();
                    if (*p == 0) break;
                } else
                    w += *p;
            }
        }
        words.push_back(fasta);
        words.push_back(alignment_path ? alignment_path : "/dev/null");
        std::vector<char*> argv;
        for (auto& w : words) argv.push_back(&w[0]);
        CParams params;
        bool expert = false;
        if (!params.parse((int)argv.size(), argv.data(), expert)) throw std::runtime_error("the reference's option parser refused: " + std::string(options ? options : ""));
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<CSequence> sequences;
        read_fasta(fasta, sequences);
        const auto t1 = std::chrono::steady_clock::now();
        Timed famsa(params);
        if (!famsa.ComputeMSA(sequences)) throw std::runtime_error("ComputeMSA failed");
        const auto t2 = std::chrono::steady_clock::now();
        times[0] = std::chrono::duration<double>(t1 - t0).count();
        times[1] = famsa.sort();
        times[2] = famsa.tree();
        times[3] = famsa.align();
        times[4] = famsa.refine();
        times[5] = std::chrono::duration<double>(t2 - t1).count();
        std::vector<CGappedSequence*> result;
        if (!famsa.GetAlignment(result)) throw std::runtime_error("no alignment");
        if (alignment_path) {
            std::ofstream out(alignment_path, std::ios::binary);
            for (CGappedSequence* g : result) out << g->id << '\n' << g->Decode() << '\n';
        }
        return (int)result.size();
    } catch (const std::exception& e) {
        if (error && error_cap > 0) {
            strncpy(error, e.what(), (size_t)error_cap - 1);
            error[error_cap - 1] = 0;
        }
        return -1;
    } catch (std::runtime_error* e) { // (the reference throws some of its errors by pointer)
        if (error && error_cap > 0) {
            strncpy(error, e->what(), (size_t)error_cap - 1);
            error[error_cap - 1] = 0;
        }
        return -1;
    }
}

} // extern "C"
