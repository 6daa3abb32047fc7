#include "seqset.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "../../include/lcsgpu.h"

namespace famsa_host {

int default_host_threads()
{
    int n = std::max(1u, std::thread::hardware_concurrency() / 2);
    // a container may grant fewer cores than the machine has (cgroup v2 cpu.max = "<quota> <period>")
    std::ifstream cg("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long period = 0;
    if (cg >> quota >> period && quota != "max" && period > 0)
        n = std::max(1, std::min(n, (int)((std::stol(quota) + period - 1) / period)));
    return n;
}

namespace {

// run fn(t) for t in [0, n_tasks) on up to n_threads threads (static split, one task per thread slot)
template <class Fn>
void parallel_for(int n_tasks, int n_threads, Fn fn)
{
    n_threads = std::max(1, std::min(n_threads, n_tasks));
    if (n_threads == 1) {
        for (int t = 0; t < n_tasks; ++t) fn(t);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::string> errors(n_threads);
    for (int w = 0; w < n_threads; ++w)
        th.emplace_back([&, w] {
            try {
                for (int t = w; t < n_tasks; t += n_threads) fn(t);
            } catch (const std::exception& e) {
                errors[w] = e.what();
            }
        });
    for (auto& t : th) t.join();
    for (const auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
}

void append_codes(Bytes& out, const char* residues, size_t n)
{
    const size_t at = out.size();
    out.resize(at + n);
    size_t m = 0;
    if (n && lcsgpu_encode(residues, n, out.data() + at, &m) != LCSGPU_OK)
        throw std::runtime_error(std::string("lcsgpu_encode: ") + lcsgpu_last_error());
    out.resize(at + m);
}

// The reader's state machine (reference core/io_service.h:99-124) over the lines of one piece of
// the file.  A piece starts at a header line (or at the start of the file), so it starts with
// the machine's clean state; only the first piece can see residue lines before any header.
// The machine runs twice over every piece with the same control flow: first into a sink that only counts (records,
// code bytes), then -- every piece knowing where its records and bytes go -- into one that encodes straight into the
// set's buffers.  (One pass into per-piece buffers and a copy afterwards touched every byte of the set twice more and
// faulted twice the memory in.)
// gaps in a residue line ('-': lcsgpu_encode drops them); memchr runs at vector speed over the usual line without any
inline size_t count_gaps(const char* p, size_t n)
{
    size_t gaps = 0;
    const char* end = p + n;
    while (p < end) {
        const char* g = (const char*)memchr(p, '-', (size_t)(end - p));
        if (!g) break;
        ++gaps;
        p = g + 1;
    }
    return gaps;
}

struct CountSink {
    size_t bytes = 0, records = 0, emitted_bytes = 0;
    void residues(const char* p, size_t n) { bytes += n - count_gaps(p, n); }
    void emit(const std::string&)
    {
        ++records;
        emitted_bytes = bytes;
    }
};
struct WriteSink {
    uint8_t* dst;      // the piece's first byte in the set's code buffer
    size_t limit;      // bytes of its records (CountSink::emitted_bytes): what lies beyond never gets a record
    uint64_t base;     // offset of dst in the code buffer
    std::string* ids;  // the piece's first record
    uint64_t* ends;    // offsets[first record + 1 ...]
    size_t at = 0, r = 0;
    void residues(const char* p, size_t n)
    {
        const size_t m = n - count_gaps(p, n);
        if (at + m <= limit) {
            size_t wrote = 0;
            if (n && lcsgpu_encode(p, n, dst + at, &wrote) != LCSGPU_OK)
                throw std::runtime_error(std::string("lcsgpu_encode: ") + lcsgpu_last_error());
        }
        at += m;
    }
    void emit(const std::string& id)
    {
        ids[r] = id;
        ends[r] = base + at;
        ++r;
    }
};

template <class Sink>
void read_piece(const char* p, const char* end, Sink& sink)
{
    std::string id;
    bool have_id = false;
    // the reference tests the RAW residue text of the pending record (io_service.h:108,122): a record
    // whose lines hold only gaps is kept, as a sequence of length 0
    bool have_residues = false;
    auto emit = [&] {
        sink.emit(id);
        have_residues = false;
    };
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* line_end = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        while (line_end > p && (line_end[-1] == '\n' || line_end[-1] == '\r')) --line_end;
        if (line_end > p) {
            if (*p == '>') {
                if (have_id && have_residues) emit();
                id.assign(p, line_end);
                have_id = true;
            } else {
                sink.residues(p, (size_t)(line_end - p));
                have_residues = true;
            }
        }
        p = next;
    }
    if (have_id && have_residues) emit();
    // residues that never got a record (no header at all, or none after them) are past the last emit: not part of the set
}

// gzip input (the reference reads .gz transparently, core/io_service.h:95): all members of the file,
// inflated into one buffer
std::vector<char> gunzip(const unsigned char* in, size_t size, const std::string& path)
{
    std::vector<char> out;
    out.resize(std::max<size_t>(size * 4, 1 << 16));
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw std::runtime_error("zlib initialisation failed");
    size_t in_pos = 0, out_pos = 0;
    for (;;) {
        if (out_pos == out.size()) out.resize(out.size() * 2);
        const size_t in_now = std::min<size_t>(size - in_pos, 1u << 30), out_now = std::min<size_t>(out.size() - out_pos, 1u << 30);
        zs.next_in = const_cast<unsigned char*>(in + in_pos);
        zs.avail_in = (uInt)in_now;
        zs.next_out = (unsigned char*)out.data() + out_pos;
        zs.avail_out = (uInt)out_now;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_pos += in_now - zs.avail_in;
        out_pos += out_now - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (in_pos >= size) break;
            if (inflateReset(&zs) != Z_OK) break; // next member
        } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
            inflateEnd(&zs);
            throw std::runtime_error("Unable to decompress input file " + path);
        } else if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_pos >= size) {
            inflateEnd(&zs);
            throw std::runtime_error("Unexpected end of compressed input file " + path);
        }
    }
    inflateEnd(&zs);
    out.resize(out_pos);
    return out;
}

} // namespace

SeqSet from_records(const std::vector<std::string>& ids, const std::vector<std::string>& residues)
{
    SeqSet s;
    s.ids = ids;
    s.offsets.assign(1, 0);
    for (const auto& r : residues) {
        append_codes(s.codes, r.data(), r.size());
        s.offsets.push_back(s.codes.size());
    }
    return s;
}

SeqSet load_fasta(const std::string& path, int n_threads)
{
    if (n_threads <= 0) n_threads = default_host_threads();
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Unable to open input file " + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        throw std::runtime_error("Unable to open input file " + path);
    }
    const size_t size = (size_t)sb.st_size;
    SeqSet s;
    s.offsets.assign(1, 0);
    if (size == 0) {
        close(fd);
        return s;
    }
    const char* mapped = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (mapped == MAP_FAILED) throw std::runtime_error("Unable to map input file " + path);
    const size_t mapped_size = size;
    std::vector<char> inflated;
    const char* buf = mapped;
    size_t size_now = size;
    if (size >= 2 && (unsigned char)mapped[0] == 0x1f && (unsigned char)mapped[1] == 0x8b) {
        try {
            inflated = gunzip((const unsigned char*)mapped, size, path);
        } catch (...) {
            munmap((void*)mapped, mapped_size);
            throw;
        }
        buf = inflated.data();
        size_now = inflated.size();
    }
    // pieces of ~equal size, each starting at a header line (a '>' right after a '\n')
    const int n_pieces = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads * 4, size_now / (1 << 20)));
    std::vector<size_t> start(n_pieces + 1, size_now);
    start[0] = 0;
    for (int k = 1; k < n_pieces; ++k) {
        size_t at = std::max(start[k - 1], size_now / n_pieces * k);
        const char* q = buf + at;
        for (;;) {
            q = (const char*)memchr(q, '\n', (size_t)(buf + size_now - q));
            if (!q || q + 1 >= buf + size_now) { at = size_now; break; }
            if (q[1] == '>') { at = (size_t)(q + 1 - buf); break; }
            ++q;
        }
        start[k] = at;
    }
    std::vector<CountSink> counted(n_pieces);
    try {
        parallel_for(n_pieces, n_threads, [&](int k) {
            if (start[k] < start[k + 1]) read_piece(buf + start[k], buf + start[k + 1], counted[k]);
        });
        std::vector<size_t> rec0(n_pieces + 1, 0), byte0(n_pieces + 1, 0);
        for (int k = 0; k < n_pieces; ++k) {
            rec0[k + 1] = rec0[k] + counted[k].records;
            byte0[k + 1] = byte0[k] + counted[k].emitted_bytes;
        }
        s.ids.resize(rec0[n_pieces]);
        s.offsets.resize(rec0[n_pieces] + 1);
        s.codes.resize(byte0[n_pieces]); // not zeroed (Bytes): every byte is written below
        parallel_for(n_pieces, n_threads, [&](int k) {
            if (start[k] >= start[k + 1]) return;
            WriteSink w{s.codes.data() + byte0[k], counted[k].emitted_bytes, byte0[k], s.ids.data() + rec0[k], s.offsets.data() + rec0[k] + 1};
            read_piece(buf + start[k], buf + start[k + 1], w);
            if (w.r != counted[k].records) throw std::runtime_error("FASTA reader: the two passes disagree");
        });
    } catch (...) {
        munmap((void*)mapped, mapped_size);
        throw;
    }
    munmap((void*)mapped, mapped_size);
    return s;
}

// The order is the reference's stable sort by (length descending, residue codes ascending), input index last.  Sorting the
// indices with a comparator that looks the sequences up costs two cache misses per comparison; here every record gets a
// 16-byte key first -- length, the first 12 residue codes as one big-endian 60-bit number (codes are < 32), the index --
// and the comparator touches the sequences only where length and prefix tie.  The index breaks the last tie, which makes a
// plain sort give the stable sort's result.  (The reference compares symbol_t = signed char; codes are < 32, so unsigned
// bytes agree.)
namespace {
struct OrderKey {
    uint64_t prefix;
    uint32_t len;
    int idx;
};
constexpr uint32_t KEY_RESIDUES = 12;
struct KeyLess {
    const SeqSet& s;
    bool operator()(const OrderKey& a, const OrderKey& b) const
    {
        if (a.len != b.len) return a.len > b.len;
        if (a.prefix != b.prefix) return a.prefix < b.prefix;
        if (a.len > KEY_RESIDUES) {
            const int c = memcmp(s.data(a.idx) + KEY_RESIDUES, s.data(b.idx) + KEY_RESIDUES, a.len - KEY_RESIDUES);
            if (c != 0) return c < 0;
        }
        return a.idx < b.idx;
    }
};
} // namespace

// order + (optionally) for every position whether its record holds the same residues as the one before it
static std::vector<int> order_and_repeats(const SeqSet& s, int n_threads, std::vector<uint8_t>* repeats)
{
    if (n_threads <= 0) n_threads = default_host_threads();
    const int n = (int)s.size();
    std::vector<OrderKey> keys(n);
    // sorted runs in parallel, then rounds of pairwise merges
    int runs = 1;
    while (runs < n_threads && n / (runs * 2) >= 4096) runs *= 2;
    std::vector<int> cut(runs + 1);
    for (int r = 0; r <= runs; ++r) cut[r] = (int)((int64_t)n * r / runs);
    const KeyLess less{s};
    parallel_for(runs, n_threads, [&](int r) {
        for (int i = cut[r]; i < cut[r + 1]; ++i) {
            const uint32_t len = s.length(i);
            const uint8_t* d = len ? (const uint8_t*)s.data(i) : nullptr;
            uint64_t prefix = 0;
            for (uint32_t k = 0; k < KEY_RESIDUES; ++k) prefix = (prefix << 5) | (k < len ? (uint64_t)(d[k] & 31) : 0u);
            keys[i] = OrderKey{prefix, len, i};
        }
        std::sort(keys.begin() + cut[r], keys.begin() + cut[r + 1], less);
    });
    std::vector<OrderKey> tmp(runs > 1 ? n : 0);
    for (int width = 1; width < runs; width *= 2) {
        const int pairs = runs / (2 * width);
        parallel_for(pairs, n_threads, [&](int q) {
            const int a = cut[2 * width * q], m = cut[2 * width * q + width], b = cut[2 * width * (q + 1)];
            std::merge(keys.begin() + a, keys.begin() + m, keys.begin() + m, keys.begin() + b, tmp.begin() + a, less);
        });
        keys.swap(tmp);
    }
    std::vector<int> order(n);
    if (repeats) repeats->assign(n, 0);
    const int slices = std::max(1, std::min(n_threads, n / 65536));
    parallel_for(slices, n_threads, [&](int t) {
        const int k0 = (int)((int64_t)n * t / slices), k1 = (int)((int64_t)n * (t + 1) / slices);
        for (int k = k0; k < k1; ++k) {
            order[k] = keys[k].idx;
            if (repeats && k > 0) {
                const OrderKey &a = keys[k], &b = keys[k - 1];
                (*repeats)[k] = a.len == b.len && a.prefix == b.prefix &&
                                (a.len <= KEY_RESIDUES || memcmp(s.data(a.idx) + KEY_RESIDUES, s.data(b.idx) + KEY_RESIDUES, a.len - KEY_RESIDUES) == 0);
            }
        }
    });
    return order;
}

std::vector<int> famsa_order(const SeqSet& s, int n_threads) { return order_and_repeats(s, n_threads, nullptr); }

WorkSet make_workset(const SeqSet& s, bool keep_duplicates, int n_threads)
{
    WorkSet w;
    std::vector<uint8_t> repeats;
    w.sorted2input = order_and_repeats(s, n_threads, keep_duplicates ? nullptr : &repeats);
    const int n = (int)w.sorted2input.size();
    w.sorted2unique.resize(n);
    int cur = -1;
    for (int k = 0; k < n; ++k) {
        const bool same = !keep_duplicates && repeats[k] != 0;
        if (!same) {
            ++cur;
            w.unique2sorted.push_back(k);
        }
        w.sorted2unique[k] = cur;
    }
    return w;
}

} // namespace famsa_host
