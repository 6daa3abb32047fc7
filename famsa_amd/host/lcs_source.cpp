#include "lcs_source.h"

#include <mutex>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "../../include/lcsgpu.h"

namespace famsa_host {

static const char* find_test_key(const char* name, size_t* len_out)
{
    const char* e = getenv("FAMSA_HOST_TEST");
    if (!e) return nullptr;
    const size_t kl = strlen(name);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        if (len >= kl && !strncmp(p, name, kl) && (len == kl || p[kl] == '=')) {
            *len_out = len;
            return p;
        }
        p += len + (end ? 1 : 0);
    }
    return nullptr;
}
bool host_test(const char* name)
{
    size_t len = 0;
    return find_test_key(name, &len) != nullptr;
}
int host_test_int(const char* key, int dflt)
{
    size_t len = 0;
    const char* p = find_test_key(key, &len);
    const size_t kl = strlen(key);
    return p && len > kl + 1 ? atoi(p + kl + 1) : dflt;
}
bool profile_on()
{
    static const bool on = getenv("LCSGPU_PROFILE") != nullptr;
    return on;
}

bool LcsSource::wide() const
{
    uint32_t m = 0;
    for (int i = 0; i < n(); ++i) m = std::max(m, length(i));
    return m > 65535;
}

void LcsSource::triangle_ids(const int* ids, int n_ids, LcsBuf& out)
{ // generic form: a rectangle, lower part kept
    out.resize(n_ids > 1 ? (size_t)n_ids * (n_ids - 1) / 2 : 0, wide());
    if (n_ids < 2) return;
    LcsBuf r;
    rect(ids, n_ids, ids, n_ids - 1, r);
    for (int i = 1; i < n_ids; ++i)
        for (int j = 0; j < i; ++j) {
            const uint32_t v = r[(size_t)i * (n_ids - 1) + j];
            const size_t k = (size_t)i * (i - 1) / 2 + j;
            if (out.wide) out.v32[k] = v; else out.v16[k] = (uint16_t)v;
        }
}

GpuLcsSource::GpuLcsSource(int device) : GpuLcsSource(std::vector<int>{device}) {}

GpuLcsSource::GpuLcsSource(const std::vector<int>& devices)
{
    if (devices.empty()) throw std::runtime_error("no GPU device given");
    for (int d : devices) {
        lcsgpu_ctx* c = nullptr;
        const int rc = lcsgpu_create(d, &c);
        if (rc != LCSGPU_OK) {
            const std::string why = lcsgpu_last_error();
            for (lcsgpu_ctx* x : ctxs_) lcsgpu_destroy(x);
            ctxs_.clear();
            throw std::runtime_error("lcsgpu_create failed (" + std::to_string(rc) + "): " + why);
        }
        ctxs_.push_back(c);
    }
    ctx_ = ctxs_[0];
}

GpuLcsSource::~GpuLcsSource()
{
    if (profile_on())
        fprintf(stderr, "engine.rect: %ld calls %.3f thread-s %.3g pairs\nengine.triangle: %ld calls %.3f thread-s %.3g pairs\n"
                        "engine.triangle_ids: %ld calls %.3f thread-s %.3g pairs\nengine.clarans: %ld calls %.3f thread-s %.3g pairs\n"
                        "engine.triangles_batch: %ld calls %.3f thread-s %.3g pairs\nengine.assign_seeds: %ld calls %.3f thread-s %.3g pairs\n",
                st_rect_.calls, st_rect_.seconds, st_rect_.pairs, st_tri_.calls, st_tri_.seconds, st_tri_.pairs,
                st_triids_.calls, st_triids_.seconds, st_triids_.pairs, st_clarans_.calls, st_clarans_.seconds,
                st_clarans_.pairs, st_batch_.calls, st_batch_.seconds, st_batch_.pairs, st_assign_.calls, st_assign_.seconds,
                st_assign_.pairs);
    for (lcsgpu_ctx* c : ctxs_) lcsgpu_destroy(c);
}

std::string GpuLcsSource::transport() const
{
    if (ctxs_.size() < 2) return std::string();
    char buf[2048];
    if (lcsgpu_multi_transport(ctxs_.data(), (int32_t)ctxs_.size(), buf, sizeof buf) != LCSGPU_OK) return std::string("unknown: ") + lcsgpu_last_error();
    return buf;
}

void GpuLcsSource::expect_threads(int n_threads)
{
    const int per_ctx = (n_threads + (int)ctxs_.size() - 1) / (int)ctxs_.size();
    for (lcsgpu_ctx* c : ctxs_) check(lcsgpu_reserve_lanes(c, per_ctx), "lcsgpu_reserve_lanes");
}

lcsgpu_ctx* GpuLcsSource::pick()
{
    if (ctxs_.size() == 1) return ctx_;
    return ctxs_[next_.fetch_add(1, std::memory_order_relaxed) % ctxs_.size()];
}

void GpuLcsSource::note(CallStat& s, double sec, double pairs)
{
    std::lock_guard<std::mutex> lk(mu_);
    s.calls++;
    s.seconds += sec;
    s.pairs += pairs;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void GpuLcsSource::check(int rc, const char* what)
{
    if (rc != LCSGPU_OK)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + lcsgpu_last_error());
}

// run fn(k) for k = 0 .. parts-1 on their own threads (part 0 on the caller's); the first error is rethrown
template <typename F>
static void on_each_device(int parts, F fn)
{
    std::vector<std::string> errors(parts);
    std::vector<std::thread> th;
    auto guarded = [&](int k) {
        try {
            fn(k);
        } catch (const std::exception& e) {
            errors[k] = e.what();
        }
    };
    for (int k = 1; k < parts; ++k) th.emplace_back(guarded, k);
    guarded(0);
    for (auto& t : th) t.join();
    for (const auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
}

void GpuLcsSource::upload(const uint8_t* codes, const std::vector<uint64_t>& offsets)
{
    upload_records(codes, offsets, nullptr, (int32_t)offsets.size() - 1);
}

void GpuLcsSource::upload_ordered(const uint8_t* codes, const std::vector<uint64_t>& offsets, const std::vector<int>& order)
{
    static_assert(sizeof(int) == sizeof(int32_t), "order entries are int32");
    upload_records(codes, offsets, order.data(), (int32_t)order.size());
}

void GpuLcsSource::upload_records(const uint8_t* codes, const std::vector<uint64_t>& offsets, const int* order, int32_t n)
{
    const int32_t n_records = (int32_t)offsets.size() - 1;
    on_each_device((int)ctxs_.size(), [&](int k) {
        const int rc = lcsgpu_upload_ordered(ctxs_[k], codes, offsets.data(), n_records, order, n);
        if (rc != LCSGPU_OK) throw std::runtime_error(std::string("lcsgpu_upload failed (") + std::to_string(rc) + "): " + lcsgpu_last_error());
    });
    lens_.resize(n);
    for (int i = 0; i < n; ++i) {
        const int r = order ? order[i] : i;
        lens_[i] = (uint32_t)(offsets[r + 1] - offsets[r]);
    }
    wide_ = LcsSource::wide();
    std::vector<uint8_t> flags(n ? n : 1);
    int32_t nq = lcsgpu_orientation_flags(ctx_, flags.data());
    if (nq < 0) check(nq, "lcsgpu_orientation_flags");
    sensitive_ = nq > 0;
}

void GpuLcsSource::add_kernel_ms(lcsgpu_ctx* ctx)
{
    double ms = 0;
    int32_t nl = 0;
    if (lcsgpu_last_kernel_ms(ctx, &ms, &nl) != LCSGPU_OK) return; // this thread's last call on that context
    std::lock_guard<std::mutex> lk(mu_);
    kernel_ms_ += ms;
}

void GpuLcsSource::triangle(int r0, int r1, LcsBuf& out)
{
    const size_t count = (size_t)r1 * (r1 - 1) / 2 - (size_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2;
    out.resize(count, wide());
    const double t0 = now_s();
    if (ctxs_.size() > 1) {
        check(lcsgpu_multi_lcs_triangle(ctxs_.data(), (int32_t)ctxs_.size(), r0, r1, out.data(), out.elem_size()),
              "lcsgpu_multi_lcs_triangle");
    } else {
        check(lcsgpu_lcs_triangle(ctx_, r0, r1, out.data(), out.elem_size()), "lcsgpu_lcs_triangle");
    }
    note(st_tri_, now_s() - t0, (double)count);
    add_kernel_ms(ctx_);
}

void GpuLcsSource::rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out)
{
    out.resize((size_t)n_refs * n_cols, wide());
    const double t0 = now_s();
    lcsgpu_ctx* c = pick();
    check(lcsgpu_lcs_rect(c, refs, 0, n_refs, cols, 0, n_cols, out.data(), n_cols, out.elem_size()),
          "lcsgpu_lcs_rect");
    note(st_rect_, now_s() - t0, (double)n_refs * n_cols);
    add_kernel_ms(c);
}

// -dist_export: unit u = slot u / n_devices of context u % n_devices, so that consecutive blocks go to different GPUs
int GpuLcsSource::text_begin(const std::vector<std::string>& ids, int distance_kind, bool square, bool pid)
{
    if (host_test("csv_host_format")) return 0; // test aid: the host formatter over lcsgpu_lcs_rect
    std::string names;
    std::vector<uint64_t> off(ids.size() + 1, 0);
    for (size_t i = 0; i < ids.size(); ++i) {
        const char* p = ids[i].c_str() + 1; // as the reference prints it: from the second character to the first NUL
        names.append(p);
        off[i + 1] = names.size();
    }
    const int slots = std::max(1, std::min(8, host_test_int("text_slots", 3)));
    const int flags = (square ? LCSGPU_TEXT_SQUARE : 0) | (pid ? LCSGPU_TEXT_PID : 0);
    for (lcsgpu_ctx* c : ctxs_) check(lcsgpu_dist_text_begin(c, names.data(), off.data(), distance_kind, flags, slots), "lcsgpu_dist_text_begin");
    text_slots_ = slots;
    return slots * (int)ctxs_.size();
}

void GpuLcsSource::text_submit(int unit, int r0, int r1)
{
    const int nd = (int)ctxs_.size();
    check(lcsgpu_dist_text_submit(ctxs_[unit % nd], unit / nd, r0, r1), "lcsgpu_dist_text_submit");
}

void GpuLcsSource::text_wait(int unit, const char*& text, uint64_t& bytes)
{
    const int nd = (int)ctxs_.size();
    const double t0 = now_s();
    check(lcsgpu_dist_text_wait(ctxs_[unit % nd], unit / nd, &text, &bytes), "lcsgpu_dist_text_wait");
    note(st_rect_, now_s() - t0, 0);
    add_kernel_ms(ctxs_[unit % nd]);
}

void GpuLcsSource::text_end()
{
    text_slots_ = 0;
    if (keep_text_buffers) return; // lcsgpu_destroy / the next upload / the next begin frees them
    for (lcsgpu_ctx* c : ctxs_) check(lcsgpu_dist_text_end(c), "lcsgpu_dist_text_end");
    text_slots_ = 0;
}

void GpuLcsSource::triangle_ids(const int* ids, int n_ids, LcsBuf& out)
{
    out.resize(n_ids > 1 ? (size_t)n_ids * (n_ids - 1) / 2 : 0, wide());
    if (n_ids < 2) return;
    const double t0 = now_s();
    lcsgpu_ctx* c = pick();
    check(lcsgpu_lcs_triangle_ids(c, ids, n_ids, out.data(), out.elem_size()), "lcsgpu_lcs_triangle_ids");
    note(st_triids_, now_s() - t0, (double)n_ids * (n_ids - 1) / 2);
    add_kernel_ms(c);
}

bool GpuLcsSource::prim_edges(int distance_kind, std::vector<MstEdge>& edges, bool triangle_orientation)
{
    static_assert(sizeof(MstEdge) == sizeof(lcsgpu_mst_edge), "edge layout");
    if (host_test("no_device_mst")) return false; // test aid: what happens when the triangle does not fit the HBM
    edges.resize(n() > 0 ? n() - 1 : 0);
    const int flags = triangle_orientation ? LCSGPU_MST_TRIANGLE_ORIENTATION : 0;
    int rc = LCSGPU_E_UNSUPPORTED;
    if (ctxs_.size() > 1)
        rc = lcsgpu_multi_mst_prim(ctxs_.data(), (int32_t)ctxs_.size(), distance_kind | flags, (lcsgpu_mst_edge*)edges.data());
    if (rc == LCSGPU_E_UNSUPPORTED) // one GPU, or an orientation-sensitive set in MSTPrim's orientation
        rc = lcsgpu_mst_prim(ctx_, distance_kind | flags, (lcsgpu_mst_edge*)edges.data());
    if (rc == LCSGPU_E_NOMEM) { // the triangle does not fit the HBM of the devices given: the caller's row-blocked host form applies
        fprintf(stderr, "[famsa-gpu] %s -- continuing with the host reduction over row blocks\n", lcsgpu_last_error());
        return false;
    }
    check(rc, "lcsgpu_mst_prim");
    add_kernel_ms(ctx_);
    return true;
}

bool GpuLcsSource::upgma_nodes(int distance_kind, bool modified, std::vector<int32_t>& left, std::vector<int32_t>& right)
{
    const int m = n() > 0 ? n() - 1 : 0;
    left.resize(m);
    right.resize(m);
    check(lcsgpu_multi_upgma(ctxs_.data(), (int32_t)ctxs_.size(), distance_kind, modified ? 1 : 0, left.data(), right.data()),
          "lcsgpu_upgma");
    add_kernel_ms(ctx_);
    return true;
}

bool GpuLcsSource::nj_nodes(int distance_kind, std::vector<int32_t>& left, std::vector<int32_t>& right)
{
    const int m = n() > 0 ? n() - 1 : 0;
    left.resize(m);
    right.resize(m);
    check(lcsgpu_multi_nj(ctxs_.data(), (int32_t)ctxs_.size(), distance_kind, left.data(), right.data()), "lcsgpu_nj");
    add_kernel_ms(ctx_);
    return true;
}

bool GpuLcsSource::triangles_batch(const int* ids, const int64_t* offsets, int n_groups, LcsBuf& out)
{
    std::vector<size_t> start((size_t)n_groups + 1, 0); // output position of every list
    for (int g = 0; g < n_groups; ++g) {
        const size_t m = (size_t)(offsets[g + 1] - offsets[g]);
        start[g + 1] = start[g] + m * (m > 0 ? m - 1 : 0) / 2;
    }
    const size_t count = start[n_groups];
    out.resize(count, wide());
    if (count == 0) return true;
    const double t0 = now_s();
    const int parts = (int)std::min<size_t>(ctxs_.size(), (size_t)n_groups);
    if (parts <= 1 || count < (1u << 16)) {
        lcsgpu_ctx* c = pick();
        check(lcsgpu_lcs_triangles_batch(c, ids, offsets, n_groups, out.data(), out.elem_size()), "lcsgpu_lcs_triangles_batch");
        add_kernel_ms(c);
    } else {
        // consecutive lists per device, cut where the running pair count passes k/parts of the total
        std::vector<int> cut(parts + 1, n_groups);
        cut[0] = 0;
        for (int k = 1; k < parts; ++k) {
            const size_t target = count / parts * k;
            int g = cut[k - 1];
            while (g < n_groups && start[g] < target) ++g;
            cut[k] = g;
        }
        on_each_device(parts, [&](int k) {
            const int g0 = cut[k], g1 = cut[k + 1];
            if (g1 <= g0) return;
            std::vector<int64_t> rel((size_t)(g1 - g0) + 1);
            for (int g = g0; g <= g1; ++g) rel[g - g0] = offsets[g] - offsets[g0];
            const int rc = lcsgpu_lcs_triangles_batch(ctxs_[k], ids + offsets[g0], rel.data(), g1 - g0,
                                                      (char*)out.data() + start[g0] * out.elem_size(), out.elem_size());
            if (rc != LCSGPU_OK) throw std::runtime_error(std::string("lcsgpu_lcs_triangles_batch failed: ") + lcsgpu_last_error());
            add_kernel_ms(ctxs_[k]);
        });
    }
    note(st_batch_, now_s() - t0, (double)count);
    return true;
}

bool GpuLcsSource::assign_seeds(const int* seeds, int n_seeds, const int* cols, int n_cols, int distance_kind, int first_k,
                                float* dist, int* assign)
{
    const double t0 = now_s();
    const int parts = (int)std::min<size_t>(ctxs_.size(), (size_t)std::max(1, n_cols / 4096));
    if (parts <= 1) {
        lcsgpu_ctx* c = pick();
        check(lcsgpu_assign_seeds(c, seeds, n_seeds, cols, n_cols, distance_kind, first_k, dist, assign), "lcsgpu_assign_seeds");
        add_kernel_ms(c);
    } else { // every column is independent of the others: the columns are split between the devices
        on_each_device(parts, [&](int k) {
            const int c0 = (int)((int64_t)n_cols * k / parts), c1 = (int)((int64_t)n_cols * (k + 1) / parts);
            if (c1 <= c0) return;
            const int rc = lcsgpu_assign_seeds(ctxs_[k], seeds, n_seeds, cols + c0, c1 - c0, distance_kind, first_k, dist + c0, assign + c0);
            if (rc != LCSGPU_OK) throw std::runtime_error(std::string("lcsgpu_assign_seeds failed: ") + lcsgpu_last_error());
            add_kernel_ms(ctxs_[k]);
        });
    }
    note(st_assign_, now_s() - t0, (double)n_seeds * n_cols);
    return true;
}

bool GpuLcsSource::clarans(const int* ids, int n_ids, int distance_kind, int n_medoids, int n_fixed,
                           float explore_fraction, int num_local, int* medoids)
{
    if (host_test("clarans_host")) return false; // the checker of the device search: keep it on the host
    const double t0 = now_s();
    lcsgpu_ctx* c = pick();
    const int rc = lcsgpu_clarans(c, ids, n_ids, distance_kind, n_medoids, n_fixed, explore_fraction, num_local, medoids);
    if (rc == LCSGPU_E_UNSUPPORTED) {
        static std::atomic<bool> told{false};
        if (profile_on() && !told.exchange(true))
            fprintf(stderr, "[famsa-gpu] %s -- such samples are searched on the host (one thread each)\n", lcsgpu_last_error());
        return false;
    }
    check(rc, "lcsgpu_clarans");
    note(st_clarans_, now_s() - t0, (double)n_ids * (n_ids - 1) / 2);
    add_kernel_ms(c);
    return true;
}

void GpuLcsSource::over_contexts(int n_jobs, const std::vector<double>& weight, const std::function<void(int, int, int)>& fn)
{
    const int nd = (int)ctxs_.size();
    if (nd == 1 || n_jobs < 2) {
        fn(0, 0, n_jobs);
        return;
    }
    double total = 0;
    for (double w : weight) total += w;
    std::vector<int> cut(1, 0);
    double acc = 0;
    for (int j = 0; j < n_jobs; ++j) {
        acc += weight[(size_t)j];
        while ((int)cut.size() < nd && acc >= total * (double)cut.size() / nd) cut.push_back(j + 1);
    }
    while ((int)cut.size() <= nd) cut.push_back(n_jobs);
    std::vector<std::thread> th;
    std::vector<std::string> err((size_t)nd);
    for (int d = 0; d < nd; ++d) {
        if (cut[(size_t)d + 1] <= cut[(size_t)d]) continue;
        th.emplace_back([&, d] {
            try {
                fn(d, cut[(size_t)d], cut[(size_t)d + 1]);
            } catch (const std::exception& e) {
                err[(size_t)d] = e.what();
            }
        });
    }
    for (auto& t : th) t.join();
    for (const auto& e : err)
        if (!e.empty()) throw std::runtime_error(e);
}

bool GpuLcsSource::clarans_batch(const int* ids, const int64_t* offsets, int n_jobs, int distance_kind, const int* n_medoids, int n_fixed,
                                 float explore_fraction, int num_local, int* medoids_out)
{
    if (host_test("clarans_host")) return false;
    const double t0 = now_s();
    std::vector<double> weight((size_t)n_jobs);
    std::vector<int64_t> med_off((size_t)n_jobs + 1, 0);
    double pairs = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const double m = (double)(offsets[j + 1] - offsets[j]);
        weight[(size_t)j] = m * m;
        pairs += m * (m - 1) / 2;
        med_off[(size_t)j + 1] = med_off[(size_t)j] + n_medoids[j];
    }
    std::atomic<bool> unsupported{false};
    over_contexts(n_jobs, weight, [&](int d, int j0, int j1) {
        std::vector<int64_t> off((size_t)(j1 - j0) + 1);
        for (int j = j0; j <= j1; ++j) off[(size_t)(j - j0)] = offsets[j] - offsets[j0];
        const int rc = lcsgpu_clarans_batch(ctxs_[(size_t)d], ids + offsets[j0], off.data(), j1 - j0, distance_kind, n_medoids + j0, n_fixed,
                                            explore_fraction, num_local, medoids_out + med_off[(size_t)j0]);
        if (rc == LCSGPU_E_UNSUPPORTED) {
            unsupported = true;
            static std::atomic<bool> told{false};
            if (profile_on() && !told.exchange(true))
                fprintf(stderr, "[famsa-gpu] %s -- such samples are searched on the host (one thread each)\n", lcsgpu_last_error());
            return;
        }
        check(rc, "lcsgpu_clarans_batch");
        add_kernel_ms(ctxs_[(size_t)d]);
    });
    if (unsupported) return false;
    note(st_clarans_, now_s() - t0, pairs);
    return true;
}

bool GpuLcsSource::assign_seeds_batch(const int* seeds, const int64_t* seed_off, const int* cols, const int64_t* col_off, int n_jobs,
                                      int distance_kind, float* dist, int* assign)
{
    const double t0 = now_s();
    std::vector<double> weight((size_t)n_jobs);
    double pairs = 0;
    for (int j = 0; j < n_jobs; ++j) pairs += weight[(size_t)j] = (double)(seed_off[j + 1] - seed_off[j]) * (double)(col_off[j + 1] - col_off[j]);
    over_contexts(n_jobs, weight, [&](int d, int j0, int j1) {
        std::vector<int64_t> so((size_t)(j1 - j0) + 1), co((size_t)(j1 - j0) + 1);
        for (int j = j0; j <= j1; ++j) {
            so[(size_t)(j - j0)] = seed_off[j] - seed_off[j0];
            co[(size_t)(j - j0)] = col_off[j] - col_off[j0];
        }
        check(lcsgpu_assign_seeds_batch(ctxs_[(size_t)d], seeds + seed_off[j0], so.data(), cols + col_off[j0], co.data(), j1 - j0, distance_kind,
                                        dist + col_off[j0], assign + col_off[j0]),
              "lcsgpu_assign_seeds_batch");
        add_kernel_ms(ctxs_[(size_t)d]);
    });
    note(st_assign_, now_s() - t0, pairs);
    return true;
}

MatrixLcsSource::MatrixLcsSource(int n, const uint32_t* lens, const uint32_t* square)
    : n_(n), lens_(lens, lens + n), m_(square, square + (size_t)n * n), sensitive_(false)
{
    wide_ = LcsSource::wide();
    for (int i = 0; i < n && !sensitive_; ++i)
        for (int j = 0; j < i; ++j)
            if (m_[(size_t)i * n + j] != m_[(size_t)j * n + i]) {
                sensitive_ = true;
                break;
            }
}

void MatrixLcsSource::triangle(int r0, int r1, LcsBuf& out)
{
    const size_t off = (size_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2;
    out.resize((size_t)r1 * (r1 - 1) / 2 - off, wide());
    for (int i = std::max(r0, 1); i < r1; ++i)
        for (int j = 0; j < i; ++j) {
            const size_t k = (size_t)i * (i - 1) / 2 + j - off;
            if (out.wide) out.v32[k] = m_[(size_t)i * n_ + j]; else out.v16[k] = (uint16_t)m_[(size_t)i * n_ + j];
        }
}

void MatrixLcsSource::rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out)
{
    out.resize((size_t)n_refs * n_cols, wide());
    for (int r = 0; r < n_refs; ++r)
        for (int c = 0; c < n_cols; ++c) {
            const uint32_t v = m_[(size_t)refs[r] * n_ + (cols ? cols[c] : c)];
            if (out.wide) out.v32[(size_t)r * n_cols + c] = v; else out.v16[(size_t)r * n_cols + c] = (uint16_t)v;
        }
}

bool MatrixLcsSource::triangles_batch(const int* ids, const int64_t* offsets, int n_groups, LcsBuf& out)
{
    size_t count = 0;
    for (int g = 0; g < n_groups; ++g) {
        const size_t m = (size_t)(offsets[g + 1] - offsets[g]);
        count += m * (m > 0 ? m - 1 : 0) / 2;
    }
    out.resize(count, wide());
    size_t k = 0;
    for (int g = 0; g < n_groups; ++g) {
        const int* list = ids + offsets[g];
        const int m = (int)(offsets[g + 1] - offsets[g]);
        for (int i = 1; i < m; ++i)
            for (int j = 0; j < i; ++j, ++k) {
                const uint32_t v = m_[(size_t)list[i] * n_ + list[j]];
                if (out.wide) out.v32[k] = v; else out.v16[k] = (uint16_t)v;
            }
    }
    return true;
}

} // namespace famsa_host
