// seqset.h -- the sequence set as the guide-tree stage sees it.
// Mirrors the state FAMSA builds before its tree generators run: FASTA records
// (reference core/io_service.h:84-127), symbol codes (core/sequence.cpp:22-80, done by
// lcsgpu_encode), the length-descending sort (msa.cpp:245-279) and duplicate removal
// (msa.cpp:338-356).  No padding is materialised: the GPU engine pads internally.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace famsa_host {

struct SeqSet {
    std::vector<std::string> ids;               // with the leading '>'
    std::vector<std::vector<uint8_t>> codes;    // symbol codes, unpadded
    size_t size() const { return ids.size(); }
    uint32_t length(size_t i) const { return (uint32_t)codes[i].size(); }
};

// FASTA reader with the reference's line handling; throws std::runtime_error on I/O errors.
SeqSet load_fasta(const std::string& path);
// Build from in-memory records (ids with '>', residue strings).
SeqSet from_records(const std::vector<std::string>& ids, const std::vector<std::string>& residues);

// Permutation `order` such that order[k] = input index of the k-th sequence in FAMSA's working
// order: stable sort by length descending, then lexicographic over the symbol codes.
std::vector<int> famsa_order(const SeqSet& s);

// Working set of the tree stage: the sorted sequences, duplicates collapsed.
struct WorkSet {
    std::vector<int> sorted2input;        // sorted position -> input index
    std::vector<int> sorted2unique;       // sorted position -> unique index ("original2mapped")
    std::vector<int> unique2sorted;       // unique index -> first sorted position holding it
    int n_sorted() const { return (int)sorted2input.size(); }
    int n_unique() const { return (int)unique2sorted.size(); }
};
WorkSet make_workset(const SeqSet& s, bool keep_duplicates);

// Concatenated codes + offsets (the lcsgpu_upload layout) of the given input indices, in order.
void pack(const SeqSet& s, const std::vector<int>& input_ids, std::vector<uint8_t>& codes,
          std::vector<uint64_t>& offsets);

} // namespace famsa_host
