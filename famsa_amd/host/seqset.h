// seqset.h -- the sequence set as the guide-tree stage sees it.
// Mirrors the state FAMSA builds before its tree generators run: FASTA records
// (reference core/io_service.h:84-127), symbol codes (core/sequence.cpp:22-80, done by
// lcsgpu_encode), the length-descending sort (msa.cpp:245-279) and duplicate removal
// (msa.cpp:338-356).  The set is held packed -- one code buffer plus offsets, the layout
// lcsgpu_upload takes -- and the reader, the sort and the gather run on all granted cores:
// at a million sequences the serial forms cost more than the whole tree stage on the GPU.
// No padding is materialised: the GPU engine pads internally.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "noinit.h"

namespace famsa_host {

// A byte buffer whose resize does not zero what it adds: the reader writes every byte itself, on all cores, and a
// std::vector<uint8_t> would first fill 765 MB (3 x 10^6 records) with zeros on one thread -- a quarter of the load stage.
using Bytes = std::vector<uint8_t, NoInit<uint8_t>>;

struct SeqSet {
    std::vector<std::string> ids;     // with the leading '>'
    Bytes codes;                      // symbol codes of all sequences, unpadded, input order
    std::vector<uint64_t> offsets;    // [size() + 1]
    size_t size() const { return ids.size(); }
    uint32_t length(size_t i) const { return (uint32_t)(offsets[i + 1] - offsets[i]); }
    const uint8_t* data(size_t i) const { return codes.data() + offsets[i]; }
};

// Worker threads for the host-side stages: the hardware threads / 2 (reference core/params.cpp:285-291),
// capped by the container's CPU quota (cgroup v2 cpu.max) when there is one.
int default_host_threads();

// FASTA reader with the reference's line handling (plain text or gzip, told apart by the magic bytes);
// throws std::runtime_error on I/O errors.
// n_threads <= 0: default_host_threads().
SeqSet load_fasta(const std::string& path, int n_threads = 0);
// Build from in-memory records (ids with '>', residue strings).
SeqSet from_records(const std::vector<std::string>& ids, const std::vector<std::string>& residues);

// Permutation `order` such that order[k] = input index of the k-th sequence in FAMSA's working
// order: stable sort by length descending, then lexicographic over the symbol codes.
std::vector<int> famsa_order(const SeqSet& s, int n_threads = 0);

// Working set of the tree stage: the sorted sequences, duplicates collapsed.
struct WorkSet {
    std::vector<int> sorted2input;        // sorted position -> input index
    std::vector<int> sorted2unique;       // sorted position -> unique index ("original2mapped")
    std::vector<int> unique2sorted;       // unique index -> first sorted position holding it
    int n_sorted() const { return (int)sorted2input.size(); }
    int n_unique() const { return (int)unique2sorted.size(); }
};
WorkSet make_workset(const SeqSet& s, bool keep_duplicates, int n_threads = 0);


} // namespace famsa_host
