#include "seqset.h"

#include <algorithm>
#include <fstream>
#include <numeric>
#include <stdexcept>

#include "../../include/lcsgpu.h"

namespace famsa_host {

static std::vector<uint8_t> encode(const std::string& residues)
{
    std::vector<uint8_t> out(residues.size() ? residues.size() : 1);
    size_t n = 0;
    if (lcsgpu_encode(residues.data(), residues.size(), out.data(), &n) != LCSGPU_OK)
        throw std::runtime_error(std::string("lcsgpu_encode: ") + lcsgpu_last_error());
    out.resize(n);
    return out;
}

SeqSet from_records(const std::vector<std::string>& ids, const std::vector<std::string>& residues)
{
    SeqSet s;
    s.ids = ids;
    s.codes.reserve(ids.size());
    for (const auto& r : residues) s.codes.push_back(encode(r));
    return s;
}

SeqSet load_fasta(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) throw std::runtime_error("Unable to open input file " + path);
    std::vector<std::string> ids, seqs;
    std::string line, id, seq;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            if (!id.empty() && !seq.empty()) { // a record needs both an id and residues
                ids.push_back(id);
                seqs.push_back(seq);
                seq.clear();
            }
            id = line;
        } else {
            seq += line;
        }
    }
    if (!id.empty() && !seq.empty()) {
        ids.push_back(id);
        seqs.push_back(seq);
    }
    return from_records(ids, seqs);
}

std::vector<int> famsa_order(const SeqSet& s)
{
    std::vector<int> order(s.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const auto &x = s.codes[a], &y = s.codes[b];
        if (x.size() != y.size()) return x.size() > y.size();
        // the reference compares symbol_t = (signed) char; codes are < 32 so unsigned agrees
        return std::lexicographical_compare(x.begin(), x.end(), y.begin(), y.end());
    });
    return order;
}

WorkSet make_workset(const SeqSet& s, bool keep_duplicates)
{
    WorkSet w;
    w.sorted2input = famsa_order(s);
    const int n = (int)w.sorted2input.size();
    w.sorted2unique.resize(n);
    int cur = -1;
    for (int k = 0; k < n; ++k) {
        const bool same = !keep_duplicates && k > 0 && s.codes[w.sorted2input[k]] == s.codes[w.sorted2input[k - 1]];
        if (!same) {
            ++cur;
            w.unique2sorted.push_back(k);
        }
        w.sorted2unique[k] = cur;
    }
    return w;
}

void pack(const SeqSet& s, const std::vector<int>& input_ids, std::vector<uint8_t>& codes,
          std::vector<uint64_t>& offsets)
{
    offsets.assign(input_ids.size() + 1, 0);
    for (size_t k = 0; k < input_ids.size(); ++k) offsets[k + 1] = offsets[k] + s.codes[input_ids[k]].size();
    codes.resize(offsets.back());
    for (size_t k = 0; k < input_ids.size(); ++k)
        std::copy(s.codes[input_ids[k]].begin(), s.codes[input_ids[k]].end(), codes.begin() + offsets[k]);
}

} // namespace famsa_host
