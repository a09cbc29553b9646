// host_baproblem.hpp -- host-side (CPU, C++) rows next to the hot path:
//   * cull() = largest_connected_component + remove_singletons + subset, iterated to a fixed point
//     (src/baproblem.rs:392-550), on the flat CSR form of vis_graph;
//   * .bal (text) and .bbal (big-endian binary) reader / writer (src/baproblem.rs:580-786).
// Irregular integer graph work and file IO: they run once per problem on the host, like in the
// reference.  Camera payloads are opaque rows of `stride` doubles (cam15 or bal9).
#pragma once
#include <algorithm>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "decimal.hpp"

namespace c2b_host {

struct Graph {
    int64_t n_cam = 0, n_pts = 0;
    int stride = 15;                 // doubles per camera row
    std::vector<double> cams, pts;   // [n_cam*stride], [n_pts*3]
    std::vector<uint64_t> row_ptr;   // [n_cam+1]
    std::vector<uint64_t> pt_idx;    // [n_obs]
    std::vector<double> uv;          // [n_obs*2]
    int64_t n_obs() const { return (int64_t)pt_idx.size(); }
};

struct UnionFind {
    std::vector<int64_t> parent, rank_;
    explicit UnionFind(int64_t n) : parent((size_t)n), rank_((size_t)n, 0) { std::iota(parent.begin(), parent.end(), 0); }
    int64_t find(int64_t x) {
        while (parent[(size_t)x] != x) {
            parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
            x = parent[(size_t)x];
        }
        return x;
    }
    void unite(int64_t a, int64_t b) {
        a = find(a); b = find(b);
        if (a == b) return;
        if (rank_[(size_t)a] < rank_[(size_t)b]) std::swap(a, b);
        parent[(size_t)b] = a;
        if (rank_[(size_t)a] == rank_[(size_t)b]) ++rank_[(size_t)a];
    }
};

// ---- cull on indices only -----------------------------------------------------------------------------------
// The fixed-point loop of cull() renumbers cameras, points and observations on every pass (and, in faithful mode,
// the observation filter depends on that numbering), but the payloads (camera rows, points, uv) only matter at the
// end.  IndexGraph carries the current numbering plus, for every camera / point / observation, where it came
// from; lcc_pass and singleton_pass are largest_connected_component (src/baproblem.rs:456-534) and
// remove_singletons (:426-453, through subset :394-423) on those indices, and the payloads are gathered once.
struct IndexGraph {
    int64_t n_cam = 0, n_pts = 0;
    std::vector<uint64_t> row_ptr, pt_idx;        // current numbering
    std::vector<int64_t> cam_orig, pt_orig;       // current -> input
    std::vector<uint64_t> edge_orig;              // current observation -> input observation
};

// keep cameras / points by mask; an observation of a kept camera survives iff edge_keep(current edge) and its point
// is kept
template <typename EdgeKeep>
inline void compact(IndexGraph &g, const std::vector<uint8_t> &keep_cam, const std::vector<uint8_t> &keep_pt, EdgeKeep edge_keep) {
    std::vector<int64_t> pt_new((size_t)g.n_pts, -1);
    int64_t np = 0;
    for (int64_t p = 0; p < g.n_pts; ++p)
        if (keep_pt[(size_t)p]) { pt_new[(size_t)p] = np; g.pt_orig[(size_t)np] = g.pt_orig[(size_t)p]; ++np; }
    g.pt_orig.resize((size_t)np);
    int64_t nc = 0;
    uint64_t w = 0, b = 0;
    for (int64_t c = 0; c < g.n_cam; ++c) {
        const uint64_t e_end = g.row_ptr[(size_t)c + 1];
        if (keep_cam[(size_t)c]) {
            g.cam_orig[(size_t)nc] = g.cam_orig[(size_t)c];
            g.row_ptr[(size_t)nc] = w;
            for (uint64_t e = b; e < e_end; ++e) {
                const int64_t p = pt_new[(size_t)g.pt_idx[(size_t)e]];
                if (p >= 0 && edge_keep(e)) { g.pt_idx[(size_t)w] = (uint64_t)p; g.edge_orig[(size_t)w] = g.edge_orig[(size_t)e]; ++w; }
            }
            ++nc;
        }
        b = e_end;
    }
    g.row_ptr[(size_t)nc] = w;
    g.row_ptr.resize((size_t)nc + 1);
    g.cam_orig.resize((size_t)nc);
    g.pt_idx.resize((size_t)w);
    g.edge_orig.resize((size_t)w);
    g.n_cam = nc;
    g.n_pts = np;
}

// largest_connected_component, src/baproblem.rs:456-534:
//   * Tie between equally large components: the reference takes HashMap iteration order
//     (nondeterministic, :483-488); here the component with the smallest member index wins.
//   * `faithful` keeps the reference's observation filter `sets[x.0] == lcc_id` (:523), which indexes the
//     camera-first union-find array with a POINT index: an observation of point j is dropped when element j of
//     that array (camera j, or point j - n_cam) lies outside the largest component.  With faithful = false the
//     filter tests the observed point itself.
inline void lcc_pass(IndexGraph &g, bool faithful) {
    if (g.n_cam == 0) return;
    const int64_t nc = g.n_cam, np = g.n_pts;
    UnionFind uf(nc + np);
    for (int64_t c = 0; c < nc; ++c)
        for (uint64_t e = g.row_ptr[(size_t)c]; e < g.row_ptr[(size_t)c + 1]; ++e) uf.unite(c, nc + (int64_t)g.pt_idx[(size_t)e]);
    std::vector<int64_t> sets((size_t)(nc + np)), canon((size_t)(nc + np), -1), size((size_t)(nc + np), 0);
    for (int64_t i = 0; i < nc + np; ++i) {
        const int64_t r = sets[(size_t)i] = uf.find(i);
        if (canon[(size_t)r] < 0) canon[(size_t)r] = i;
        ++size[(size_t)r];
    }
    int64_t lcc = -1, best = -1, best_canon = -1;
    for (int64_t r = 0; r < nc + np; ++r)
        if (size[(size_t)r] > 0 && (size[(size_t)r] > best || (size[(size_t)r] == best && canon[(size_t)r] < best_canon))) {
            best = size[(size_t)r]; lcc = r; best_canon = canon[(size_t)r];
        }
    std::vector<uint8_t> keep_cam((size_t)nc), keep_pt((size_t)np);
    for (int64_t c = 0; c < nc; ++c) keep_cam[(size_t)c] = sets[(size_t)c] == lcc;
    for (int64_t p = 0; p < np; ++p) keep_pt[(size_t)p] = sets[(size_t)(nc + p)] == lcc;
    if (faithful) {
        // the reference's filter (:523) looks up element `point index` of the camera-first array
        const std::vector<uint64_t> old_pt(g.pt_idx);               // compact() overwrites pt_idx in place
        compact(g, keep_cam, keep_pt, [&](uint64_t e) { return sets[(size_t)old_pt[(size_t)e]] == lcc; });
    } else {
        compact(g, keep_cam, keep_pt, [](uint64_t) { return true; });
    }
}

// remove_singletons, src/baproblem.rs:426-453: cameras need > 3 observations, points > 1; the point counts are taken
// over ALL cameras, including the ones being removed (reference TODO at :437).
inline void singleton_pass(IndexGraph &g) {
    std::vector<uint8_t> keep_cam((size_t)g.n_cam), keep_pt((size_t)g.n_pts);
    for (int64_t c = 0; c < g.n_cam; ++c) keep_cam[(size_t)c] = g.row_ptr[(size_t)c + 1] - g.row_ptr[(size_t)c] > 3;
    std::vector<int64_t> count((size_t)g.n_pts, 0);
    for (uint64_t p : g.pt_idx) ++count[(size_t)p];
    for (int64_t p = 0; p < g.n_pts; ++p) keep_pt[(size_t)p] = count[(size_t)p] > 1;
    compact(g, keep_cam, keep_pt, [](uint64_t) { return true; });
}

// cull, src/baproblem.rs:538-549 (mode 0); mode 1: largest_connected_component once; mode 2: remove_singletons once
inline Graph cull(const Graph &g, bool faithful, int mode = 0) {
    IndexGraph ig;
    ig.n_cam = g.n_cam; ig.n_pts = g.n_pts;
    ig.row_ptr = g.row_ptr; ig.pt_idx = g.pt_idx;
    ig.cam_orig.resize((size_t)g.n_cam); std::iota(ig.cam_orig.begin(), ig.cam_orig.end(), (int64_t)0);
    ig.pt_orig.resize((size_t)g.n_pts); std::iota(ig.pt_orig.begin(), ig.pt_orig.end(), (int64_t)0);
    ig.edge_orig.resize(g.pt_idx.size()); std::iota(ig.edge_orig.begin(), ig.edge_orig.end(), (uint64_t)0);
    int64_t nc = g.n_cam, np = g.n_pts;
    if (mode != 2) lcc_pass(ig, faithful);
    if (mode != 1) singleton_pass(ig);
    while (mode == 0 && (ig.n_cam != nc || ig.n_pts != np)) {
        nc = ig.n_cam; np = ig.n_pts;
        lcc_pass(ig, faithful);
        singleton_pass(ig);
    }
    Graph o;
    o.stride = g.stride;
    o.n_cam = ig.n_cam; o.n_pts = ig.n_pts;
    o.cams.resize((size_t)o.n_cam * g.stride);
    for (int64_t c = 0; c < o.n_cam; ++c)
        std::copy(&g.cams[(size_t)ig.cam_orig[(size_t)c] * g.stride], &g.cams[(size_t)(ig.cam_orig[(size_t)c] + 1) * g.stride],
                  &o.cams[(size_t)c * g.stride]);
    o.pts.resize((size_t)o.n_pts * 3);
    for (int64_t p = 0; p < o.n_pts; ++p)
        std::copy(&g.pts[(size_t)ig.pt_orig[(size_t)p] * 3], &g.pts[(size_t)ig.pt_orig[(size_t)p] * 3 + 3], &o.pts[(size_t)p * 3]);
    o.row_ptr = ig.row_ptr;
    if (o.row_ptr.empty()) o.row_ptr.assign(1, 0);
    o.pt_idx = ig.pt_idx;
    o.uv.resize(2 * ig.edge_orig.size());
    for (size_t e = 0; e < ig.edge_orig.size(); ++e) {
        o.uv[2 * e] = g.uv[2 * (size_t)ig.edge_orig[e]];
        o.uv[2 * e + 1] = g.uv[2 * (size_t)ig.edge_orig[e] + 1];
    }
    return o;
}

// ---- .bal / .bbal -----------------------------------------------------------------------------------
// fn(0) ... fn(n - 1), one per thread (fn(0) on the caller), all joined before it returns.  Starting a thread can fail
// (std::system_error, EAGAIN under a container's pids limit): the indices that got no thread then run on the caller, and
// the threads that did start are joined whatever happens -- unwinding past a joinable std::thread is std::terminate, which
// would take the embedding process down where the C ABI promises a status code.
template <typename F>
inline void run_threads(int n, F &&fn) {
    std::vector<std::thread> th;
    struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } join_all{th};
    th.reserve(n > 1 ? (size_t)(n - 1) : 0);
    int started = 1;
    for (; started < n; ++started) {
        try { th.emplace_back([&fn, started]() { fn(started); }); }
        catch (const std::system_error &) { break; }
    }
    if (n > 0) fn(0);
    for (int k = started; k < n; ++k) fn(k);
}

// threads of the text formatter / parser: what the calling problem's options ask for (c2b_problem_options.io_threads,
// in force on this thread for the duration of its call), else the process-wide setting (c2b_host_set_io_threads), else the
// usable cores, at most 16.  No environment variable: a caller of the C ABI sets these through the ABI.
inline std::atomic<int> &io_threads_setting() { static std::atomic<int> v{0}; return v; }
inline int &io_threads_of_this_call() { static thread_local int v = 0; return v; }
struct IoThreadsScope {                              // RAII: a problem's own thread count while one of its calls runs
    int saved;
    explicit IoThreadsScope(int n) : saved(io_threads_of_this_call()) { if (n > 0) io_threads_of_this_call() = n; }
    ~IoThreadsScope() { io_threads_of_this_call() = saved; }
};
inline int io_threads() {
    int v = io_threads_of_this_call();
    if (v < 1) v = io_threads_setting().load(std::memory_order_relaxed);
    if (v >= 1) return std::min(v, 64);
    return (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
}

// Rust's `{}` for f64: shortest digits that round-trip, never an exponent, "1" for 1.0, "-0", "NaN", "inf".
// (decimal.hpp -- the code the device writer runs; std::to_chars(fixed) prints the exact integer above 2^53 instead.)
inline void fmt_f64(double v, std::string &out) {
    const c2b_dec::Text t = c2b_dec::describe(v, &c2b_dec::host_tables());
    const size_t at = out.size();
    out.resize(at + t.len);
    c2b_dec::emit(t, &out[at]);
}

// write_text, src/baproblem.rs:709-733 (cameras: 9 values on ONE line, space-joined).  Shortest round-trip decimals are
// the cost (170 MB/s on one thread: 7 s for the 1.2 GB of `synthetic --blocks 128`), so the lines are formatted by a
// pool of threads in tasks of kTextTask lines -- observation lines, then camera lines, then point lines -- while this
// thread writes the finished tasks in order (and frees them): the file is byte for byte the sequential writer's.
constexpr int64_t kTextTask = 1 << 17;
inline bool write_text(const char *path, const Graph &g, std::string *err, int n_threads = 0) {
    if (g.stride != 9) { *err = "write_text needs bal9 camera rows"; return false; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { *err = std::string("cannot create ") + path; return false; }
    const int64_t n_obs = g.n_obs();
    const int64_t t_obs = (n_obs + kTextTask - 1) / kTextTask, t_cam = (g.n_cam + kTextTask - 1) / kTextTask,
                  t_pts = (g.n_pts + kTextTask - 1) / kTextTask, n_tasks = t_obs + t_cam + t_pts;
    auto format_task = [&](int64_t k, std::string &s) {
        s.clear();
        if (k < t_obs) {
            const int64_t a = k * kTextTask, b = std::min(n_obs, a + kTextTask);
            s.reserve((size_t)(b - a) * 52);
            // the camera whose list holds observation a, then walk
            int64_t c = (int64_t)(std::upper_bound(g.row_ptr.begin(), g.row_ptr.end(), (uint64_t)a) - g.row_ptr.begin()) - 1;
            for (int64_t e = a; e < b; ++e) {
                while ((uint64_t)e >= g.row_ptr[(size_t)c + 1]) ++c;
                s += std::to_string(c); s += ' ';
                s += std::to_string(g.pt_idx[(size_t)e]); s += ' ';
                fmt_f64(g.uv[2 * (size_t)e], s); s += ' ';
                fmt_f64(g.uv[2 * (size_t)e + 1], s); s += '\n';
            }
        } else if (k < t_obs + t_cam) {
            const int64_t a = (k - t_obs) * kTextTask, b = std::min(g.n_cam, a + kTextTask);
            for (int64_t c = a; c < b; ++c) {
                for (int q = 0; q < 9; ++q) { if (q) s += ' '; fmt_f64(g.cams[(size_t)c * 9 + q], s); }
                s += '\n';
            }
        } else {
            const int64_t a = (k - t_obs - t_cam) * kTextTask, b = std::min(g.n_pts, a + kTextTask);
            for (int64_t p = a; p < b; ++p) {
                fmt_f64(g.pts[(size_t)p * 3], s); s += ' ';
                fmt_f64(g.pts[(size_t)p * 3 + 1], s); s += ' ';
                fmt_f64(g.pts[(size_t)p * 3 + 2], s); s += '\n';
            }
        }
    };
    bool ok = true;
    {
        const std::string head = std::to_string(g.n_cam) + " " + std::to_string(g.n_pts) + " " + std::to_string(n_obs) + "\n";
        ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
    }
    if (n_threads <= 0) n_threads = io_threads();
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_tasks));
    if (n_threads <= 1) {
        std::string s;
        for (int64_t k = 0; k < n_tasks && ok; ++k) { format_task(k, s); ok = std::fwrite(s.data(), 1, s.size(), f) == s.size(); }
    } else {
        // at most 4 * n_threads formatted tasks wait for the writer at any time (a few hundred MB at worst)
        std::vector<std::string> out((size_t)n_tasks);
        std::vector<char> done((size_t)n_tasks, 0);
        std::mutex mu;
        std::condition_variable cv;
        int64_t next = 0, written = 0;
        bool stop = false;
        auto worker = [&]() {
            while (true) {
                int64_t k;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || next >= n_tasks || next < written + 4 * n_threads; });
                    if (stop || next >= n_tasks) return;
                    k = next++;
                }
                std::string s;
                format_task(k, s);
                std::lock_guard<std::mutex> lk(mu);
                out[(size_t)k] = std::move(s);
                done[(size_t)k] = 1;
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        // whatever happens below, the workers are told to stop and joined (a joinable thread must never be unwound past)
        struct StopAndJoin {
            std::vector<std::thread> &v; std::mutex &m; std::condition_variable &c; bool &stop;
            ~StopAndJoin() { { std::lock_guard<std::mutex> lk(m); stop = true; } c.notify_all(); for (auto &t : v) if (t.joinable()) t.join(); }
        } stop_and_join{pool, mu, cv, stop};
        try { for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker); }
        catch (const std::system_error &) { }               // fewer workers than asked for; none at all: this thread formats
        if (pool.empty()) {
            std::string s;
            for (int64_t k = 0; k < n_tasks && ok; ++k) { format_task(k, s); ok = std::fwrite(s.data(), 1, s.size(), f) == s.size(); }
        } else for (int64_t k = 0; k < n_tasks; ++k) {
            std::string s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done[(size_t)k] != 0; });
                s = std::move(out[(size_t)k]);
            }
            if (ok) ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
            std::lock_guard<std::mutex> lk(mu);
            written = k + 1;
            cv.notify_all();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) *err = std::string("write failed: ") + path;
    return ok;
}

inline void fmt_f32(float v, std::string &out) {
    if (v != v) { out += "NaN"; return; }
    char buf[128];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    out.append(buf, r.ptr);
}

// write_cameras, src/bin/city2ba.rs:359-439 (ply-rs ASCII writer: header, then one line per element with the
// properties in declaration order, space separated)
inline bool write_ply(const char *path, int64_t n_cam, const double *centers, int64_t n_pts, const double *pts,
                      const uint64_t *row_ptr, const uint64_t *pt_idx, std::string *err) {
    FILE *f = std::fopen(path, "wb");
    if (!f) { *err = std::string("cannot create ") + path; return false; }
    const uint64_t n_obs = n_cam ? row_ptr[(size_t)n_cam] : 0;
    std::string s;
    s.reserve(1 << 20);
    s += "ply\nformat ascii 1.0\nelement vertex " + std::to_string(n_cam + n_pts) + "\n";
    s += "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n";
    s += "element edge " + std::to_string(n_obs) + "\nproperty int vertex1\nproperty int vertex2\nend_header\n";
    auto flush = [&](bool force) {
        if (force || s.size() > (1u << 20)) { std::fwrite(s.data(), 1, s.size(), f); s.clear(); }
    };
    auto vertex = [&](const double *p, const char *rgb) {
        for (int k = 0; k < 3; ++k) { fmt_f32((float)p[k], s); s += ' '; }
        s += rgb;
        flush(false);
    };
    for (int64_t c = 0; c < n_cam; ++c) vertex(centers + 3 * c, "255 0 0\n");
    for (int64_t p = 0; p < n_pts; ++p) vertex(pts + 3 * p, "0 255 0\n");
    for (int64_t c = 0; c < n_cam; ++c)
        for (uint64_t e = row_ptr[(size_t)c]; e < row_ptr[(size_t)c + 1]; ++e) {
            s += std::to_string((int32_t)c); s += ' ';
            s += std::to_string((int32_t)(pt_idx[(size_t)e] + (uint64_t)n_cam)); s += '\n';
            flush(false);
        }
    flush(true);
    const bool ok = std::fclose(f) == 0;
    if (!ok) *err = std::string("write failed: ") + path;
    return ok;
}

inline void put_be64(std::string &s, uint64_t v) {
    char b[8];
    for (int i = 0; i < 8; ++i) b[i] = (char)(v >> (56 - 8 * i));
    s.append(b, 8);
}
inline void put_bef64(std::string &s, double d) {
    uint64_t v;
    std::memcpy(&v, &d, 8);
    put_be64(s, v);
}

// write_binary, src/baproblem.rs:736-764
inline bool write_binary(const char *path, const Graph &g, std::string *err) {
    if (g.stride != 9) { *err = "write_binary needs bal9 camera rows"; return false; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { *err = std::string("cannot create ") + path; return false; }
    std::string s;
    put_be64(s, (uint64_t)g.n_cam); put_be64(s, (uint64_t)g.n_pts); put_be64(s, (uint64_t)g.n_obs());
    auto flush = [&](bool force) {
        if (force || s.size() > (1u << 20)) { std::fwrite(s.data(), 1, s.size(), f); s.clear(); }
    };
    for (int64_t c = 0; c < g.n_cam; ++c) {
        put_be64(s, g.row_ptr[(size_t)c + 1] - g.row_ptr[(size_t)c]);
        for (uint64_t e = g.row_ptr[(size_t)c]; e < g.row_ptr[(size_t)c + 1]; ++e) {
            put_be64(s, g.pt_idx[(size_t)e]);
            put_bef64(s, g.uv[2 * (size_t)e]);
            put_bef64(s, g.uv[2 * (size_t)e + 1]);
        }
        flush(false);
    }
    for (double v : g.cams) { put_bef64(s, v); flush(false); }
    for (double v : g.pts) { put_bef64(s, v); flush(false); }
    flush(true);
    const bool ok = std::fclose(f) == 0;
    if (!ok) *err = std::string("write failed: ") + path;
    return ok;
}

inline bool read_all(const char *path, std::string &buf, std::string *err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { *err = std::string("cannot open ") + path; return false; }
    if (std::fseek(f, 0, SEEK_END) == 0) {                   // sized in one go when the file says how long it is
        const long n = std::ftell(f);
        std::rewind(f);
        if (n > 0) {
            buf.resize((size_t)n);
            const size_t got = std::fread(&buf[0], 1, (size_t)n, f);
            buf.resize(got);
        }
    }
    char tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return true;
}

// from_file_text, src/baproblem.rs:580-628: unsigned/float tokens separated by any whitespace; observations
// are (camera, point, u, v) tuples in file order and are pushed per camera like BAProblem::new (:342-355).
inline bool read_text_sequential(const std::string &buf, Graph &g, std::string *err) {
    const char *p = buf.c_str(), *end = p + buf.size();
    auto skip = [&]() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; };
    auto get_u = [&](uint64_t &v) -> bool {
        skip();
        if (p >= end || *p < '0' || *p > '9') return false;
        v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            if (v > (UINT64_MAX - 9) / 10) return false;         // would wrap: not a count or an index
            v = v * 10 + (uint64_t)(*p++ - '0');
        }
        return true;
    };
    auto get_d = [&](double &v) -> bool {
        skip();
        if (p >= end) return false;
        char *q = nullptr;
        v = std::strtod(p, &q);
        if (q == p) return false;
        p = q;
        return true;
    };
    uint64_t nc, np, no;
    if (!get_u(nc) || !get_u(np) || !get_u(no)) { *err = "ParseError: bad header"; return false; }
    // The header is untrusted: every observation takes at least 8 bytes of text ("0 0 0 0\n"), every camera 18, every
    // point 6, so counts beyond what the rest of the file can hold are rejected before anything is sized from them.
    const uint64_t left = (uint64_t)(end - p);
    if (no > left / 8 || nc > left / 18 || np > left / 6) {
        *err = "ParseError: header counts (" + std::to_string(nc) + " cameras, " + std::to_string(np) + " points, " +
               std::to_string(no) + " observations) exceed what a file of this size can hold";
        return false;
    }
    std::vector<uint64_t> oc((size_t)no), op((size_t)no);
    std::vector<double> ou((size_t)no * 2);
    for (uint64_t i = 0; i < no; ++i)
        if (!get_u(oc[(size_t)i]) || !get_u(op[(size_t)i]) || !get_d(ou[2 * (size_t)i]) || !get_d(ou[2 * (size_t)i + 1])) {
            *err = "ParseError: bad observation " + std::to_string(i);
            return false;
        }
    g.stride = 9;
    g.n_cam = (int64_t)nc; g.n_pts = (int64_t)np;
    g.cams.resize((size_t)nc * 9);
    g.pts.resize((size_t)np * 3);
    for (double &v : g.cams) if (!get_d(v)) { *err = "ParseError: bad camera block"; return false; }
    for (double &v : g.pts) if (!get_d(v)) { *err = "ParseError: bad point block"; return false; }
    // BAProblem::new: asserts then per-camera push in file order
    std::vector<uint64_t> count((size_t)nc + 1, 0);
    for (uint64_t i = 0; i < no; ++i) {
        if (oc[(size_t)i] >= nc) { *err = "assertion failed: cam_i < cams.len()"; return false; }
        if (op[(size_t)i] >= np) { *err = "assertion failed: p_i < points.len()"; return false; }
        ++count[(size_t)oc[(size_t)i] + 1];
    }
    g.row_ptr.assign((size_t)nc + 1, 0);
    for (uint64_t c = 0; c < nc; ++c) g.row_ptr[(size_t)c + 1] = g.row_ptr[(size_t)c] + count[(size_t)c + 1];
    std::vector<uint64_t> fill(g.row_ptr.begin(), g.row_ptr.end() - 1);
    g.pt_idx.resize((size_t)no);
    g.uv.resize((size_t)no * 2);
    for (uint64_t i = 0; i < no; ++i) {
        const uint64_t d = fill[(size_t)oc[(size_t)i]]++;
        g.pt_idx[(size_t)d] = op[(size_t)i];
        g.uv[2 * (size_t)d] = ou[2 * (size_t)i];
        g.uv[2 * (size_t)d + 1] = ou[2 * (size_t)i + 1];
    }
    return true;
}

// The same parse on several threads, for files whose numbers are separated by whitespace (every file this writer or the
// reference's writer produces; nom's grammar also accepts numbers glued to each other, e.g. "1.5-3").  The token stream
// is positional -- 3 header tokens, 4 per observation, 9 per camera, 3 per point -- so: cut the buffer into chunks at
// whitespace, count each chunk's tokens, prefix-sum, and let every thread parse its tokens knowing their global index.
// Anything irregular (a token a number parser does not consume entirely, too few tokens, a spelling only strtod takes)
// makes the whole call fall back to read_text_sequential, which owns the error messages and the corner cases.
// 12.7 s -> ~1 s for the 1.2 GB text form of `synthetic --blocks 128` on 16 threads.
inline bool read_text(const char *path, Graph &g, std::string *err, int n_threads = 0) {
    std::string buf;
    if (!read_all(path, buf, err)) return false;
    if (n_threads <= 0) n_threads = io_threads();
    const size_t size = buf.size();
    if (n_threads <= 1 || size < ((size_t)1 << 20)) return read_text_sequential(buf, g, err);
    const char *base = buf.c_str();
    auto is_ws = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r'; };
    const int T = n_threads;
    std::vector<size_t> cut((size_t)T + 1, size);
    cut[0] = 0;
    for (int k = 1; k < T; ++k) {
        size_t q = size / (size_t)T * (size_t)k;
        if (q < cut[(size_t)k - 1]) q = cut[(size_t)k - 1];
        while (q < size && !is_ws(base[q])) ++q;             // a chunk never starts inside a token
        cut[(size_t)k] = q;
    }
    std::vector<uint64_t> n_tok((size_t)T, 0);
    auto for_tokens = [&](int k, auto &&fn) {                // fn(first, last) for every token starting in chunk k
        const char *p = base + cut[(size_t)k], *e = base + cut[(size_t)k + 1];
        while (true) {
            while (p < e && is_ws(*p)) ++p;
            if (p >= e) return;
            const char *q = p;
            while (q < base + size && !is_ws(*q)) ++q;      // (a token that starts in the chunk ends in it: cuts are at whitespace)
            fn(p, q);
            p = q;
        }
    };
    run_threads(T, [&](int k) { uint64_t c = 0; for_tokens(k, [&](const char *, const char *) { ++c; }); n_tok[(size_t)k] = c; });
    std::vector<uint64_t> tok0((size_t)T + 1, 0);
    for (int k = 0; k < T; ++k) tok0[(size_t)k + 1] = tok0[(size_t)k] + n_tok[(size_t)k];
    // header: the first three tokens, wherever they are
    uint64_t hdr[3] = {0, 0, 0};
    {
        int got = 0;
        bool bad = false;
        for (int k = 0; k < T && got < 3 && !bad; ++k)
            for_tokens(k, [&](const char *a, const char *b) {
                if (got >= 3 || bad) return;
                uint64_t v = 0;
                const auto r = std::from_chars(a, b, v);
                if (r.ec != std::errc() || r.ptr != b) { bad = true; return; }
                hdr[got++] = v;
            });
        if (bad || got < 3) return read_text_sequential(buf, g, err);
    }
    const uint64_t nc = hdr[0], np = hdr[1], no = hdr[2];
    if (no > size / 8 || nc > size / 18 || np > size / 6) return read_text_sequential(buf, g, err);   // it words the error
    const uint64_t need = 3 + 4 * no + 9 * nc + 3 * np;
    if (tok0[(size_t)T] < need) return read_text_sequential(buf, g, err);
    std::vector<uint64_t> oc((size_t)no), op((size_t)no);
    std::vector<double> ou((size_t)no * 2);
    g.stride = 9;
    g.n_cam = (int64_t)nc; g.n_pts = (int64_t)np;
    g.cams.resize((size_t)nc * 9);
    g.pts.resize((size_t)np * 3);
    std::atomic<int> irregular{0};
    const uint64_t obs_end = 3 + 4 * no, cam_end = obs_end + 9 * nc;
    {
        run_threads(T, [&](int k) {
                uint64_t gi = tok0[(size_t)k];
                for_tokens(k, [&](const char *a, const char *b) {
                    const uint64_t i = gi++;
                    if (i < 3 || i >= need || irregular.load(std::memory_order_relaxed)) return;
                    if (i < obs_end) {
                        const uint64_t o = (i - 3) >> 2, fld = (i - 3) & 3;
                        if (fld < 2) {
                            uint64_t v = 0;
                            const auto r = std::from_chars(a, b, v);
                            if (r.ec != std::errc() || r.ptr != b) { irregular = 1; return; }
                            (fld == 0 ? oc : op)[(size_t)o] = v;
                            return;
                        }
                        // strtod, like the sequential parser (libstdc++ 11's from_chars<double> is strtod behind a locale
                        // switch and a lock: it got slower with every thread added)
                        char *q = nullptr;
                        ou[2 * (size_t)o + (fld - 2)] = std::strtod(a, &q);
                        if (q != b) irregular = 1;
                        return;
                    }
                    double &dst = i < cam_end ? g.cams[(size_t)(i - obs_end)] : g.pts[(size_t)(i - cam_end)];
                    char *q = nullptr;
                    dst = std::strtod(a, &q);
                    if (q != b) irregular = 1;
                });
            });
    }
    if (irregular) { g = Graph(); return read_text_sequential(buf, g, err); }
    // BAProblem::new: asserts then per-camera push in file order
    std::vector<uint64_t> count((size_t)nc + 1, 0);
    bool sorted = true;
    for (uint64_t i = 0; i < no; ++i) {
        if (oc[(size_t)i] >= nc) { *err = "assertion failed: cam_i < cams.len()"; return false; }
        if (op[(size_t)i] >= np) { *err = "assertion failed: p_i < points.len()"; return false; }
        if (i && oc[(size_t)i] < oc[(size_t)i - 1]) sorted = false;
        ++count[(size_t)oc[(size_t)i] + 1];
    }
    g.row_ptr.assign((size_t)nc + 1, 0);
    for (uint64_t c = 0; c < nc; ++c) g.row_ptr[(size_t)c + 1] = g.row_ptr[(size_t)c] + count[(size_t)c + 1];
    if (sorted) {                                            // camera-major already (every writer's order): nothing moves
        g.pt_idx = std::move(op);
        g.uv = std::move(ou);
        return true;
    }
    std::vector<uint64_t> fill(g.row_ptr.begin(), g.row_ptr.end() - 1);
    g.pt_idx.resize((size_t)no);
    g.uv.resize((size_t)no * 2);
    for (uint64_t i = 0; i < no; ++i) {
        const uint64_t d = fill[(size_t)oc[(size_t)i]]++;
        g.pt_idx[(size_t)d] = op[(size_t)i];
        g.uv[2 * (size_t)d] = ou[2 * (size_t)i];
        g.uv[2 * (size_t)d + 1] = ou[2 * (size_t)i + 1];
    }
    return true;
}

// from_file_binary, src/baproblem.rs:632-693
inline bool read_binary(const char *path, Graph &g, std::string *err) {
    std::string buf;
    if (!read_all(path, buf, err)) return false;
    size_t pos = 0;
    auto get64 = [&](uint64_t &v) -> bool {
        if (pos + 8 > buf.size()) return false;
        v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | (uint8_t)buf[pos + i];
        pos += 8;
        return true;
    };
    auto getf = [&](double &d) -> bool {
        uint64_t v;
        if (!get64(v)) return false;
        std::memcpy(&d, &v, 8);
        return true;
    };
    uint64_t nc, np, no;
    if (!get64(nc) || !get64(np) || !get64(no)) { *err = "Binary parse error"; return false; }
    // untrusted header: a camera costs 8 + 72 bytes, a point 24, an observation 24 -- reject counts the file cannot hold
    const uint64_t left = (uint64_t)(buf.size() - pos);
    if (nc > left / 80 || np > left / 24 || no > left / 24) { *err = "Binary parse error: header counts exceed the file size"; return false; }
    g.stride = 9;
    g.n_cam = (int64_t)nc; g.n_pts = (int64_t)np;
    g.row_ptr.assign(1, 0);
    for (uint64_t c = 0; c < nc; ++c) {
        uint64_t k;
        if (!get64(k)) { *err = "Binary parse error"; return false; }
        for (uint64_t i = 0; i < k; ++i) {
            uint64_t p; double u, v;
            if (!get64(p) || !getf(u) || !getf(v)) { *err = "Binary parse error"; return false; }
            g.pt_idx.push_back(p); g.uv.push_back(u); g.uv.push_back(v);
        }
        g.row_ptr.push_back((uint64_t)g.pt_idx.size());
    }
    g.cams.resize((size_t)nc * 9);
    g.pts.resize((size_t)np * 3);
    for (double &v : g.cams) if (!getf(v)) { *err = "Binary parse error"; return false; }
    for (double &v : g.pts) if (!getf(v)) { *err = "Binary parse error"; return false; }
    for (uint64_t p : g.pt_idx) if (p >= np) { *err = "Binary parse error: point index out of range"; return false; }
    return true;
}

inline std::string extension(const char *path) {
    const std::string s(path);
    const size_t dot = s.find_last_of('.'), slash = s.find_last_of('/');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
    return s.substr(dot + 1);
}

}  // namespace c2b_host
