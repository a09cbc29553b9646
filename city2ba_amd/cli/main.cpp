// city2ba (MI355X build) -- command line with the reference's subcommands and flag names
// (src/bin/city2ba.rs:113-260), written against the C ABI only (include/city2ba_hip.h), i.e. exactly
// what a Rust host would call.
//
//   city2ba synthetic OUT [--blocks N --cameras-per-block N --points-per-block N --max-dist X
//                          --camera-height X --point-height X --block-inset X --block-length X]
//   city2ba synthetic-line OUT [--cameras N --points N --max-dist X --camera-height X --point-height X
//                               --point-offset X --length X]
//   city2ba noise IN OUT [--rotation-std X --translation-std X --point-std X --observation-std X
//                         --drift-std X --drift-strength X --fixed-drift --drift-angle X
//                         --sin-strength X --sin-frequency X --mismatch-chance X --drop-features X
//                         --split-landmarks X --join-landmarks X] [--seed N] [--gpus N | --devices a,b,...]
//
//   city2ba generate FILE OUT [--cameras N --intrinsics-start x,y,z --intrinsics-end x,y,z --points N --max-dist X
//                              --ground X --height X --no-lcc --move-to-origin --path NAME --step-size X] [--seed N]
//                             [--exact-lcc]   (extension: cull without the reference's observation-filter quirk at
//                                              src/baproblem.rs:523, which discards most observations whenever the
//                                              graph holds unseen points)
//   city2ba ply IN OUT
//
// `generate` casts its rays by brute force over the triangles instead of through Embree.  Every random draw is
// seeded (--seed; default: std::random_device) where the reference uses thread_rng().
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/city2ba_hip.h"
#include "../../include/city2ba_hip_host.h"
#include "../../include/city2ba_hip_experimental.h"

namespace {

// Leaves at once: std::exit would run static destructors (the HIP runtime's among them) while a helper thread -- the
// mesh loader, the hierarchy builder of run_generate -- may still be running.
[[noreturn]] void die(const std::string &msg) {
    std::fflush(stdout);
    std::fprintf(stderr, "Error: %s\n", msg.c_str());
    std::fflush(stderr);
    std::_Exit(1);
}

void ck(int rc) {
    if (rc != C2B_OK) die(c2b_last_error());
}

// Rust `{:.2e}`: "1.23e-5", "0.00e0"
std::string sci2(double v) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    if (v == 0.0) return "0.00e0";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.2e", v);
    std::string s(buf);
    const size_t e = s.find('e');
    const int ex = std::atoi(s.c_str() + e + 1);
    return s.substr(0, e) + "e" + std::to_string(ex);
}

struct Args {
    std::vector<std::string> positional;
    std::map<std::string, std::string> opt;
    bool has(const std::string &k) const { return opt.count(k) != 0; }
    double f(const std::string &k, double dflt) const {
        auto it = opt.find(k);
        if (it == opt.end()) return dflt;
        char *end = nullptr;
        const double v = std::strtod(it->second.c_str(), &end);
        if (end == it->second.c_str() || *end) die("Invalid value for '--" + k + " <" + k + ">': invalid float literal");
        return v;
    }
    int64_t i(const std::string &k, int64_t dflt) const {
        auto it = opt.find(k);
        if (it == opt.end()) return dflt;
        char *end = nullptr;
        const long long v = std::strtoll(it->second.c_str(), &end, 10);
        if (end == it->second.c_str() || *end || v < 0) die("Invalid value for '--" + k + " <" + k + ">': invalid digit found in string");
        return (int64_t)v;
    }
};

Args parse(int argc, char **argv, int first, const std::vector<std::string> &flags, const std::vector<std::string> &values) {
    Args a;
    for (int k = first; k < argc; ++k) {
        std::string s = argv[k];
        if (s.rfind("--", 0) == 0) {
            std::string key = s.substr(2), val;
            const size_t eq = key.find('=');
            bool has_val = false;
            if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); has_val = true; }
            bool is_flag = false, is_value = false;
            for (auto &f : flags) if (f == key) is_flag = true;
            for (auto &v : values) if (v == key) is_value = true;
            if (is_flag) { a.opt[key] = "1"; continue; }
            if (!is_value) die("Found argument '--" + key + "' which wasn't expected, or isn't valid in this context");
            if (!has_val) {
                if (k + 1 >= argc) die("The argument '--" + key + " <" + key + ">' requires a value but none was supplied");
                val = argv[++k];
            }
            a.opt[key] = val;
        } else {
            a.positional.push_back(s);
        }
    }
    return a;
}

// The library takes its behaviour switches as c2b_problem_options, not from the environment; this program -- a caller of
// the C ABI like any other -- keeps its environment variables (README.md) and turns them into those options for every
// problem it makes.
static int env_int(const char *name, int dflt) { const char *v = std::getenv(name); return v && *v ? std::atoi(v) : dflt; }
static bool env_on(const char *name) { const char *v = std::getenv(name); return v && *v && std::strcmp(v, "0") != 0; }
static int create_problem(int device, c2b_problem **out) {
    const int rc = c2b_problem_create(device, out);
    if (rc) return rc;
    c2b_problem_options o;
    c2b_problem_options_init(&o);
    o.host_text = env_on("C2B_HOST_TEXT");
    o.text_device_strict = env_on("C2B_TEXT_DEVICE_STRICT");
    o.read_threads = std::max(0, std::min(64, env_int("C2B_READ_THREADS", 0)));
    o.io_threads = std::max(0, std::min(64, env_int("C2B_IO_THREADS", 0)));
    o.rank_sort_max_row = std::max(0, env_int("C2B_RANK_SORT_MAX_ROW", 0));
    if (const char *v = std::getenv("C2B_TEXT_DEVICE_MIN_BYTES")) o.text_device_min_bytes = (int64_t)std::strtoll(v, nullptr, 10);
    return c2b_problem_set_options(*out, &o);
}

// C2B_TIMING=1: wall time of each phase on stderr
struct PhaseTimer {
    bool on = std::getenv("C2B_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[timing] %-50s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

struct HostProblem {
    int64_t n_cam = 0, n_pts = 0;
    std::vector<double> cams15, pts, uv;
    std::vector<uint64_t> row_ptr, pt_idx;
};

// the resident problem (cameras as in-memory records, points, graph, observations) on the host
void download_problem(c2b_problem *p, HostProblem &hp) {
    int64_t n_obs = 0;
    ck(c2b_problem_sizes(p, &hp.n_cam, &hp.n_pts, &n_obs));
    hp.cams15.assign((size_t)hp.n_cam * 15 + 1, 0.0);
    hp.pts.assign((size_t)hp.n_pts * 3 + 1, 0.0);
    hp.uv.assign((size_t)n_obs * 2 + 1, 0.0);
    hp.row_ptr.assign((size_t)hp.n_cam + 1, 0);
    hp.pt_idx.assign((size_t)n_obs + 1, 0);
    ck(c2b_problem_download(p, hp.cams15.data(), hp.pts.data(), hp.uv.data()));
    ck(c2b_problem_download_graph(p, hp.row_ptr.data(), hp.pt_idx.data()));
    hp.cams15.resize((size_t)hp.n_cam * 15);
    hp.pts.resize((size_t)hp.n_pts * 3);
    hp.uv.resize((size_t)n_obs * 2);
    hp.pt_idx.resize((size_t)n_obs);
}

void display_and_write(c2b_problem *p, const HostProblem &hp, const std::string &out) {
    PhaseTimer timer;
    std::printf("Bundle Adjustment Problem with %lld cameras, %lld points, and %lld observations\n", (long long)hp.n_cam,
                (long long)hp.n_pts, (long long)hp.pt_idx.size());
    std::vector<double> bal9((size_t)hp.n_cam * 9);
    ck(c2b_problem_download_bal(p, bal9.data()));
    timer.mark("to_vec + download");
    ck(c2b_bal_write(out.c_str(), hp.n_cam, bal9.data(), hp.n_pts, hp.pts.data(), hp.row_ptr.data(), hp.pt_idx.data(), hp.uv.data()));
    timer.mark("write");
}

// synthetic_grid / synthetic_line (src/synthetic.rs:163-300, :313-381) and the tail of their subcommands
// (src/bin/city2ba.rs:447-478), everything on the resident problem: the layout loops, the visibility loop -- candidates
// within max_dist, hits_building, the predicate --, cull, the Display line, and the file image.  Only the file's bytes
// leave the device.  C2B_HOST_CANDIDATES=1 takes rounds 1-3's route (layout and candidate search on the host, 47.5 M
// pairs uploaded at --blocks 128, the arrays downloaded and serialised by the host) for comparison; same file.
struct Layout {
    bool grid = true;
    int64_t cpb = 0, ppb = 0, blocks = 0, n_cam = 0, n_pts = 0;              // grid: per block; line: totals in n_cam / n_pts
    double L = 0, inset = 0, cam_h = 0, pt_h = 0, length = 0, point_offset = 0;
};

void generate_cull_write(c2b_problem *p, const Layout &lay, double max_dist, const std::string &out, PhaseTimer &timer) {
    const bool host_route = std::getenv("C2B_HOST_CANDIDATES") != nullptr;
    const bool occlusion = lay.grid;
    const double L = lay.grid ? lay.L : 1.0, inset = lay.grid ? lay.inset : 0.0;
    if (host_route) {
        int64_t n_cam = lay.n_cam, n_pts = lay.n_pts;
        if (lay.grid) ck(c2b_synthetic_grid_sizes(lay.cpb, lay.ppb, lay.blocks, &n_cam, &n_pts));
        std::vector<double> pos((size_t)n_cam * 3), dir((size_t)n_cam * 9), pts((size_t)n_pts * 3), cams15((size_t)n_cam * 15);
        if (lay.grid) ck(c2b_synthetic_grid_layout(lay.cpb, lay.ppb, lay.blocks, lay.L, lay.inset, lay.cam_h, lay.pt_h, pos.data(), dir.data(), pts.data()));
        else ck(c2b_synthetic_line_layout(n_cam, n_pts, lay.length, lay.point_offset, lay.cam_h, lay.pt_h, pos.data(), dir.data(), pts.data()));
        ck(c2b_problem_from_position_direction(p, n_cam, pos.data(), dir.data(), cams15.data()));
        std::vector<uint64_t> empty_rows((size_t)n_cam + 1, 0);
        ck(c2b_problem_upload(p, n_cam, cams15.data(), n_pts, pts.data(), empty_rows.data(), nullptr, nullptr));
        timer.mark("layout (host) + from_position_direction + upload");
        std::vector<double> centers((size_t)n_cam * 3);
        ck(c2b_problem_centers(p, centers.data()));
        c2b_pairs *pairs = nullptr;
        const int threads = (int)std::max(1u, std::thread::hardware_concurrency());
        ck(c2b_candidate_pairs(centers.data(), n_cam, pts.data(), n_pts, max_dist, 0, n_cam, occlusion ? 1 : 0, L, inset, threads, &pairs));
        timer.mark("centres + candidate pairs (host, threaded)");
        std::vector<uint64_t> rows((size_t)n_cam + 1, 0);
        ck(c2b_problem_visibility_pairs_compact(p, c2b_pairs_count(pairs), c2b_pairs_cam_idx(pairs), c2b_pairs_pt_idx(pairs), max_dist, rows.data()));
        c2b_pairs_free(pairs);
        timer.mark("visibility predicate + compaction (device)");
    } else {
        if (lay.grid) ck(c2b_problem_synthetic_grid_layout(p, lay.cpb, lay.ppb, lay.blocks, lay.L, lay.inset, lay.cam_h, lay.pt_h));
        else ck(c2b_problem_synthetic_line_layout(p, lay.n_cam, lay.n_pts, lay.length, lay.point_offset, lay.cam_h, lay.pt_h));
        timer.mark("layout (device)");
        ck(c2b_problem_visibility_within_distance(p, max_dist, occlusion ? 1 : 0, L, inset, nullptr));
        timer.mark("candidates + hits_building + predicate (device)");
    }
    ck(c2b_problem_adopt_visibility(p));
    // .cull(), src/synthetic.rs:299 -- on the device
    ck(c2b_problem_cull(p, 1));
    timer.mark("adopt + cull (device)");
    if (host_route) {
        HostProblem hp;
        download_problem(p, hp);
        timer.mark("download");
        display_and_write(p, hp, out);
        return;
    }
    int64_t nc = 0, np = 0, no = 0;
    ck(c2b_problem_sizes(p, &nc, &np, &no));
    std::printf("Bundle Adjustment Problem with %lld cameras, %lld points, and %lld observations\n", (long long)nc, (long long)np, (long long)no);
    ck(c2b_problem_write(p, out.c_str(), -1));
    timer.mark("write (c2b_problem_write: the file image is built on the device)");
}

int run_synthetic(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {}, {"cameras-per-block", "points-per-block", "max-dist", "camera-height",
                                             "point-height", "block-inset", "block-length", "blocks", "device"});
    if (a.positional.size() != 1) die("The following required arguments were not provided:\n    <OUTPUT>");
    Layout lay;
    lay.cpb = a.i("cameras-per-block", 10); lay.ppb = a.i("points-per-block", 10); lay.blocks = a.i("blocks", 5);
    lay.cam_h = a.f("camera-height", 1); lay.pt_h = a.f("point-height", 1);
    lay.inset = a.f("block-inset", 1); lay.L = a.f("block-length", 20);
    PhaseTimer timer;
    c2b_problem *p = nullptr;
    ck(create_problem((int)a.i("device", 0), &p));
    timer.mark("problem_create (HIP runtime start)");
    generate_cull_write(p, lay, a.f("max-dist", 10), a.positional[0], timer);
    c2b_problem_destroy(p);
    return 0;
}

int run_synthetic_line(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {}, {"cameras", "points", "max-dist", "camera-height", "point-height",
                                             "point-offset", "length", "device"});
    if (a.positional.size() != 1) die("The following required arguments were not provided:\n    <OUTPUT>");
    Layout lay;
    lay.grid = false;
    lay.n_cam = a.i("cameras", 10); lay.n_pts = a.i("points", 10);
    lay.length = a.f("length", 20); lay.point_offset = a.f("point-offset", 1);
    lay.cam_h = a.f("camera-height", 1); lay.pt_h = a.f("point-height", 1);
    PhaseTimer timer;
    c2b_problem *p = nullptr;
    ck(create_problem((int)a.i("device", 0), &p));
    timer.mark("problem_create (HIP runtime start)");
    generate_cull_write(p, lay, a.f("max-dist", 10), a.positional[0], timer);
    c2b_problem_destroy(p);
    return 0;
}

// `noise --gpus N` (or --devices a,b,...): run_noise (src/bin/city2ba.rs:280-357) with the problem sharded over N GPUs of
// this node, driven from ONE process -- SURVEY 8(b)'s ctx_create(n_dev, dev_ids).  The index-corruption passes and their
// culls run on the host arrays as in the single-GPU path; then cameras are cut into N contiguous ranges holding equal
// numbers of observations (c2b_partition_cameras), every GPU gets its range, that range's observations and ALL points
// (one c2b_problem each, marked with c2b_problem_set_shard), and one host thread per GPU runs the same sequence of
// collective Level-1 calls (c2b_problem_*_sharded: statistics and errors go through RCCL, c2b_comm_init_all).  Draws
// are keyed by global indices, so the written file is the same for every N up to rounding: the statistics that scale
// the perturbations (mean, std) are sums of per-rank shares, whose last bits depend on how the cameras were cut.
int run_noise_sharded(const Args &a, uint64_t seed, int64_t n_cam, int64_t n_pts, int64_t n_obs, std::vector<double> &bal9,
                      std::vector<double> &pts, size_t pts_cap, std::vector<double> &uv, std::vector<uint64_t> &row_ptr,
                      std::vector<uint64_t> &pt_idx) {
    std::vector<int> devs;
    if (a.has("devices")) {
        const std::string &sv = a.opt.at("devices");
        size_t pos = 0;
        while (pos <= sv.size()) {
            const size_t comma = std::min(sv.find(',', pos), sv.size());
            const std::string tok = sv.substr(pos, comma - pos);
            char *end = nullptr;
            const long v = std::strtol(tok.c_str(), &end, 10);
            if (tok.empty() || *end || v < 0) die("Invalid value for '--devices <devices>': expected a,b,c");
            devs.push_back((int)v);
            pos = comma + 1;
        }
    } else {
        for (int64_t k = 0; k < a.i("gpus", 1); ++k) devs.push_back((int)k);
    }
    const int N = (int)devs.size();
    if (N < 1 || N > 64) die("Invalid value for '--gpus <gpus>': expected 1..64");
    int visible = 0;
    ck(c2b_device_count(&visible));
    for (int d : devs) if (d >= visible) die("--gpus / --devices: device " + std::to_string(d) + " is not visible (" + std::to_string(visible) + " devices)");

    std::vector<c2b_comm *> comms((size_t)N, nullptr);
    ck(c2b_comm_init_all(N, devs.data(), comms.data()));
    double err[4] = {0, 0, 0, 0};                                         // initial L1, L2, final L1, L2 (written by rank 0)
    std::vector<double> pts_out;

    // One pass over the GPUs: cut the cameras, give every GPU its shard, run the collective sequence on one host thread
    // per GPU.  with_initial: evaluate the error of the problem as uploaded; with_noise: drift, sine, Gaussian noise, final
    // error, download.
    auto pass = [&](bool with_initial, bool with_noise) {
        std::vector<int64_t> bounds((size_t)N + 1);
        ck(c2b_partition_cameras(row_ptr.data(), n_cam, N, bounds.data()));
        int nonempty = 0;
        for (int k = 0; k < N; ++k) nonempty += bounds[(size_t)k + 1] > bounds[(size_t)k];
        if (nonempty < N)
            die("--gpus " + std::to_string(N) + ": the problem's " + std::to_string(n_cam) + " cameras (" + std::to_string(n_obs) +
                " observations) cut into only " + std::to_string(nonempty) + " non-empty ranges; use --gpus " + std::to_string(std::max(1, nonempty)));
        std::vector<std::string> failures((size_t)N);
        if (with_noise) {
            std::fprintf(stderr, "noise on %d GPU%s through %s; observations per GPU:", N, N == 1 ? "" : "s", c2b_comm_backend());
            for (int k = 0; k < N; ++k)
                std::fprintf(stderr, " %llu", (unsigned long long)(row_ptr[(size_t)bounds[(size_t)k + 1]] - row_ptr[(size_t)bounds[(size_t)k]]));
            std::fprintf(stderr, "\n");
            pts_out.assign((size_t)n_pts * 3 + 1, 0.0);
        }
        std::vector<std::thread> workers;
        for (int k = 0; k < N; ++k) {
            workers.emplace_back([&, k]() {
                // a failing call ends THIS worker with its message (reported after the join; std::exit from a worker thread
                // would run static destructors under the other workers' feet).  A rank that fails between two collectives
                // leaves its peers waiting in the next one -- the same as a rank of an MPI job dying
                auto ck = [&](int rc) { if (rc != C2B_OK) throw std::runtime_error(c2b_last_error()); };
                try {
                const int64_t lo = bounds[(size_t)k], hi = bounds[(size_t)k + 1], nc = hi - lo;
                const uint64_t o0 = row_ptr[(size_t)lo];
                std::vector<uint64_t> rp((size_t)nc + 1);
                for (int64_t c = 0; c <= nc; ++c) rp[(size_t)c] = row_ptr[(size_t)(lo + c)] - o0;
                c2b_problem *p = nullptr;
                ck(create_problem(devs[(size_t)k], &p));
                ck(c2b_problem_upload_bal(p, nc, bal9.data() + 9 * lo, n_pts, pts.data(), rp.data(), pt_idx.data() + o0, uv.data() + 2 * o0));
                ck(c2b_problem_set_shard(p, lo, n_cam, (int64_t)o0));
                c2b_comm *comm = comms[(size_t)k];
                double l1, l2;
                if (with_initial) {           // both norms from one pass, ONE 2-element all-reduce (src/bin/city2ba.rs:283-287)
                    ck(c2b_problem_total_reprojection_errors_l1_l2_sharded(p, comm, &l1, &l2));
                    if (k == 0) { err[0] = l1; err[1] = l2; }
                }
                if (with_noise) {
                    // src/bin/city2ba.rs:305-316: drift is ALWAYS applied, even with zero strength
                    if (a.has("fixed-drift")) {
                        double stats[C2B_STATS_DOUBLES];
                        ck(c2b_problem_stats_sharded(p, comm, stats));
                        ck(c2b_problem_add_drift_sharded(p, comm, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), stats + 3, seed));
                    } else {
                        ck(c2b_problem_add_drift_sharded(p, comm, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), nullptr, seed));
                    }
                    if (a.f("sin-strength", 0) > 0.0) {        // :318-333
                        const double dx[3] = {1, 0, 0}, dz[3] = {0, 0, 1}, up[3] = {0, 1, 0};
                        ck(c2b_problem_add_sin_noise_sharded(p, comm, dx, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
                        ck(c2b_problem_add_sin_noise_sharded(p, comm, dz, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
                    }
                    // :334-340 (always) and :350-354: the observation pass and both final errors in one launch per shard
                    ck(c2b_problem_add_noise_errors_l1_l2_sharded(p, comm, a.f("translation-std", 0), a.f("rotation-std", 0),
                                                                  a.f("point-std", 0), a.f("observation-std", 0), seed + 1, &l1, &l2));
                    if (k == 0) { err[2] = l1; err[3] = l2; }
                    // every shard's cameras and observations go back to their rows of the host arrays; the points
                    // (identical on every GPU) come from rank 0
                    ck(c2b_problem_download(p, nullptr, k == 0 ? pts_out.data() : nullptr, uv.data() + 2 * o0));
                    ck(c2b_problem_download_bal(p, bal9.data() + 9 * lo));
                }
                c2b_problem_destroy(p);
                } catch (const std::exception &e) { failures[(size_t)k] = e.what(); }
            });
        }
        for (auto &w : workers) w.join();
        for (int k = 0; k < N; ++k)
            if (!failures[(size_t)k].empty()) die("GPU " + std::to_string(devs[(size_t)k]) + " (rank " + std::to_string(k) + "): " + failures[(size_t)k]);
    };

    // host-side index corruption + cull, exactly as in the single-GPU path (src/bin/city2ba.rs:288-303): the reference
    // prints the initial error BEFORE these passes, so when any of them is requested the problem visits the GPUs twice
    const double split = a.f("split-landmarks", 0.0);
    const bool reshapes = a.f("drop-features", 1.0) < 1.0 || a.f("join-landmarks", 0.0) > 0.0 || split > 0.0;
    if (reshapes) pass(true, false);
    auto cull = [&]() {
        ck(c2b_cull(&n_cam, bal9.data(), 9, &n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data(), 1));
        n_obs = (int64_t)row_ptr[(size_t)n_cam];
    };
    if (a.f("drop-features", 1.0) < 1.0) {
        ck(c2b_drop_features(n_cam, row_ptr.data(), pt_idx.data(), uv.data(), a.f("drop-features", 1.0), seed + 2));
        cull();
    }
    if (a.f("join-landmarks", 0.0) > 0.0) {
        ck(c2b_join_landmarks(n_pts, pts.data(), (int64_t)row_ptr[(size_t)n_cam], pt_idx.data(), split, seed + 3));
        cull();
    }
    if (split > 0.0) {
        ck(c2b_split_landmarks(&n_pts, pts.data(), (int64_t)pts_cap, (int64_t)row_ptr[(size_t)n_cam], pt_idx.data(), split, seed + 4));
        cull();
    }
    if (n_cam == 0 || n_pts == 0) die("EmptyProblem: nothing remains after culling");
    pass(!reshapes, true);
    for (c2b_comm *c : comms) c2b_comm_destroy(c);
    std::copy(pts_out.begin(), pts_out.begin() + (size_t)n_pts * 3, pts.begin());
    std::printf("Initial error: %s (L1) %s (L2)\n", sci2(err[0]).c_str(), sci2(err[1]).c_str());
    if (a.f("mismatch-chance", 0.0) > 0.0)                               // :341, on the noised image positions (host arrays)
        ck(c2b_add_incorrect_correspondences(n_cam, row_ptr.data(), pt_idx.data(), uv.data(), a.f("mismatch-chance", 0.0), seed + 5));
    std::printf("BA Problem with %lld cameras, %lld points, %lld correspondences\n", (long long)n_cam, (long long)n_pts,
                (long long)n_obs);
    if (a.f("mismatch-chance", 0.0) > 0.0) {
        // the reference reports the error AFTER the correspondences were scrambled: one more sharded pass would need a
        // re-upload; the single-GPU evaluation of the final arrays gives the same number
        c2b_problem *p = nullptr;
        ck(create_problem(devs[0], &p));
        ck(c2b_problem_upload_bal(p, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
        ck(c2b_problem_total_reprojection_errors_l1_l2(p, &err[2], &err[3]));
        c2b_problem_destroy(p);
    }
    std::printf("Final error: %s (L1) %s (L2)\n", sci2(err[2]).c_str(), sci2(err[3]).c_str());
    ck(c2b_bal_write(a.positional[1].c_str(), n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    return 0;
}

int run_noise(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {"fixed-drift"},
                         {"rotation-std", "translation-std", "point-std", "observation-std", "drift-std", "drift-strength",
                          "drift-angle", "mismatch-chance", "drop-features", "split-landmarks", "join-landmarks",
                          "sin-strength", "sin-frequency", "seed", "device", "gpus", "devices"});
    if (a.positional.size() != 2) die("The following required arguments were not provided:\n    <FILE> <OUT>");
    uint64_t seed;
    if (a.has("seed")) seed = (uint64_t)a.i("seed", 0);
    else { std::random_device rd; seed = ((uint64_t)rd() << 32) ^ rd(); }   // the reference: unseeded thread_rng()

    // The common case -- no index-corruption pass (src/bin/city2ba.rs:288-303, :341), one GPU -- never leaves the device:
    // the file is decoded into the resident problem (c2b_problem_read), the whole chain of src/bin/city2ba.rs:280-357 runs
    // on it, and the output file's image is assembled there (c2b_problem_write).  Both error pairs cost one pass each, the
    // second one fused with add_noise's observation pass.
    const bool reshapes_any = a.f("drop-features", 1.0) < 1.0 || a.f("join-landmarks", 0.0) > 0.0 || a.f("split-landmarks", 0.0) > 0.0;
    if (!reshapes_any && !(a.f("mismatch-chance", 0.0) > 0.0) && !a.has("gpus") && !a.has("devices") && !std::getenv("C2B_HOST_IO")) {
        PhaseTimer timer;
        c2b_problem *p = nullptr;
        ck(create_problem((int)a.i("device", 0), &p));
        timer.mark("problem_create (HIP runtime start)");
        ck(c2b_problem_read(p, a.positional[0].c_str(), -1));
        timer.mark("read (c2b_problem_read: decoded on the device)");
        double l1, l2;
        ck(c2b_problem_total_reprojection_errors_l1_l2(p, &l1, &l2));
        std::printf("Initial error: %s (L1) %s (L2)\n", sci2(l1).c_str(), sci2(l2).c_str());
        if (a.has("fixed-drift")) {                                      // :305-316: drift is ALWAYS applied
            double stats[C2B_STATS_DOUBLES];
            ck(c2b_problem_stats(p, stats));
            ck(c2b_problem_add_drift(p, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), stats + 3, seed));
        } else {
            ck(c2b_problem_add_drift_normalized(p, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), seed));
        }
        if (a.f("sin-strength", 0) > 0.0) {                              // :318-333
            const double dx[3] = {1, 0, 0}, dz[3] = {0, 0, 1}, up[3] = {0, 1, 0};
            ck(c2b_problem_add_sin_noise(p, dx, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
            ck(c2b_problem_add_sin_noise(p, dz, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
        }
        ck(c2b_problem_add_noise_errors_l1_l2(p, a.f("translation-std", 0), a.f("rotation-std", 0), a.f("point-std", 0),
                                              a.f("observation-std", 0), seed + 1, &l1, &l2));      // :334-340 + :350-354
        int64_t nc = 0, np = 0, no = 0;
        ck(c2b_problem_sizes(p, &nc, &np, &no));
        std::printf("BA Problem with %lld cameras, %lld points, %lld correspondences\n", (long long)nc, (long long)np, (long long)no);
        std::printf("Final error: %s (L1) %s (L2)\n", sci2(l1).c_str(), sci2(l2).c_str());
        timer.mark("errors + drift + noise (device)");
        ck(c2b_problem_write(p, a.positional[1].c_str(), -1));
        timer.mark("write (c2b_problem_write: the file image is built on the device)");
        c2b_problem_destroy(p);
        return 0;
    }

    c2b_balfile *f = nullptr;
    ck(c2b_bal_read(a.positional[0].c_str(), &f));
    int64_t n_cam, n_pts, n_obs;
    ck(c2b_bal_sizes(f, &n_cam, &n_pts, &n_obs));
    const double split = a.f("split-landmarks", 0.0);
    // room for the landmarks split_landmarks appends
    const size_t pts_cap = (size_t)n_pts + (split > 0.0 ? (size_t)(std::min(split, 1.0) * (double)n_pts) + 1 : 0);
    std::vector<double> bal9((size_t)n_cam * 9 + 1), pts(pts_cap * 3 + 1), uv((size_t)n_obs * 2 + 1);
    std::vector<uint64_t> row_ptr((size_t)n_cam + 1), pt_idx((size_t)n_obs + 1);
    ck(c2b_bal_copy(f, bal9.data(), pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    c2b_bal_close(f);

    if (a.has("gpus") || a.has("devices"))
        return run_noise_sharded(a, seed, n_cam, n_pts, n_obs, bal9, pts, pts_cap, uv, row_ptr, pt_idx);

    c2b_problem *p = nullptr;
    ck(create_problem((int)a.i("device", 0), &p));
    ck(c2b_problem_upload_bal(p, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    double l1, l2;
    ck(c2b_problem_total_reprojection_errors_l1_l2(p, &l1, &l2));        // src/bin/city2ba.rs:283-287: both norms, one pass
    std::printf("Initial error: %s (L1) %s (L2)\n", sci2(l1).c_str(), sci2(l2).c_str());

    // index-corruption passes on the host arrays, each followed by cull() (src/bin/city2ba.rs:288-303); camera rows
    // stay 9-vectors (cull treats them as opaque)
    bool reshaped = false;
    auto cull = [&]() {
        ck(c2b_cull(&n_cam, bal9.data(), 9, &n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data(), 1));
        n_obs = (int64_t)row_ptr[(size_t)n_cam];
        reshaped = true;
    };
    if (a.f("drop-features", 1.0) < 1.0) {
        ck(c2b_drop_features(n_cam, row_ptr.data(), pt_idx.data(), uv.data(), a.f("drop-features", 1.0), seed + 2));
        cull();
    }
    if (a.f("join-landmarks", 0.0) > 0.0) {
        // the reference passes opt.split_landmarks as the fraction here (src/bin/city2ba.rs:296)
        ck(c2b_join_landmarks(n_pts, pts.data(), (int64_t)row_ptr[(size_t)n_cam], pt_idx.data(), split, seed + 3));
        cull();
    }
    if (split > 0.0) {
        ck(c2b_split_landmarks(&n_pts, pts.data(), (int64_t)pts_cap, (int64_t)row_ptr[(size_t)n_cam], pt_idx.data(), split, seed + 4));
        cull();
    }
    if (reshaped) {
        if (n_cam == 0 || n_pts == 0) die("EmptyProblem: nothing remains after culling");
        ck(c2b_problem_upload_bal(p, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    }

    // src/bin/city2ba.rs:305-316: drift is ALWAYS applied, even with zero strength
    if (a.has("fixed-drift")) {
        double stats[C2B_STATS_DOUBLES];
        ck(c2b_problem_stats(p, stats));
        ck(c2b_problem_add_drift(p, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), stats + 3, seed));
    } else {
        ck(c2b_problem_add_drift_normalized(p, a.f("drift-strength", 0), a.f("drift-angle", 0), a.f("drift-std", 0), seed));
    }
    if (a.f("sin-strength", 0) > 0.0) {        // :318-333
        const double dx[3] = {1, 0, 0}, dz[3] = {0, 0, 1}, up[3] = {0, 1, 0};
        ck(c2b_problem_add_sin_noise(p, dx, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
        ck(c2b_problem_add_sin_noise(p, dz, up, a.f("sin-strength", 0), a.f("sin-frequency", 1)));
    }
    // :334-340 (always) fused with the final errors of :350-354: the observation pass draws, perturbs, stores, projects
    // and folds both norms in one launch.  With --mismatch-chance the errors are re-evaluated after the scramble below.
    ck(c2b_problem_add_noise_errors_l1_l2(p, a.f("translation-std", 0), a.f("rotation-std", 0), a.f("point-std", 0),
                                          a.f("observation-std", 0), seed + 1, &l1, &l2));
    ck(c2b_problem_download(p, nullptr, pts.data(), uv.data()));
    if (a.f("mismatch-chance", 0.0) > 0.0) {                             // :341, on the noised image positions
        std::vector<double> cams15((size_t)n_cam * 15 + 1);
        ck(c2b_problem_download(p, cams15.data(), nullptr, nullptr));
        ck(c2b_add_incorrect_correspondences(n_cam, row_ptr.data(), pt_idx.data(), uv.data(), a.f("mismatch-chance", 0.0), seed + 5));
        ck(c2b_problem_upload(p, n_cam, cams15.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
        ck(c2b_problem_total_reprojection_errors_l1_l2(p, &l1, &l2));
    }
    std::printf("BA Problem with %lld cameras, %lld points, %lld correspondences\n", (long long)n_cam, (long long)n_pts,
                (long long)n_obs);
    std::printf("Final error: %s (L1) %s (L2)\n", sci2(l1).c_str(), sci2(l2).c_str());
    ck(c2b_problem_download_bal(p, bal9.data()));
    ck(c2b_bal_write(a.positional[1].c_str(), n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    c2b_problem_destroy(p);
    return 0;
}

// parse_vec3, src/bin/city2ba.rs:20-31: "x,y,z"
void parse_vec3(const Args &a, const std::string &k, double out[3]) {
    auto it = a.opt.find(k);
    if (it == a.opt.end()) return;
    const std::string &s = it->second;
    size_t pos = 0;
    for (int c = 0; c < 3; ++c) {
        const size_t comma = c < 2 ? s.find(',', pos) : s.size();
        if (comma == std::string::npos) die("Invalid value for '--" + k + " <" + k + ">': expected x,y,z");
        const std::string tok = s.substr(pos, comma - pos);
        char *end = nullptr;
        out[c] = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str() || *end) die("Invalid value for '--" + k + " <" + k + ">': invalid float literal");
        pos = comma + 1;
    }
}

// Rust `{}` of an f64: shortest digits that round-trip, never scientific
std::string display_f64(double v) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

// run_generate, src/bin/city2ba.rs:480-573.  Ray casts: brute force over the mesh's triangles (host for camera
// placement, the device occlusion kernel for the visibility graph) in place of Embree.
int run_generate(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {"no-lcc", "move-to-origin", "exact-lcc"},
                         {"cameras", "intrinsics-start", "intrinsics-end", "points", "max-dist", "ground", "height", "path",
                          "step-size", "seed", "device"});
    if (a.positional.size() != 2) die("The following required arguments were not provided:\n    <FILE> <OUT>");
    if (a.has("path") && a.has("ground")) die("The argument '--path <path>' cannot be used with '--ground <ground>'");
    const int64_t num_cameras = a.i("cameras", 100), num_points = a.i("points", 1000);
    const double max_dist = a.f("max-dist", 100), ground = a.f("ground", 0), height = a.f("height", 1);
    const double step_size = a.f("step-size", 0);
    double istart[3] = {1, 0, 0}, iend[3] = {1, 0, 0};
    parse_vec3(a, "intrinsics-start", istart);
    parse_vec3(a, "intrinsics-end", iend);
    uint64_t seed;
    if (a.has("seed")) seed = (uint64_t)a.i("seed", 0);
    else { std::random_device rd; seed = ((uint64_t)rd() << 32) ^ rd(); }   // the reference: unseeded thread_rng()

    PhaseTimer timer;
    // Everything that needs only the file runs on a second thread while this one starts the HIP runtime (c2b_problem_create:
    // 60-260 ms that need nothing from the file) and then works on the cameras: load the mesh, find the path, move to
    // the origin, extract the triangles -- "mesh ready" -- and, when the cameras follow a path, go straight on to the
    // hierarchy of the occlusion rays (c2b_bvh_build, itself multi-threaded), which is not needed before the points are
    // sampled and the candidates found.  (Poisson placement builds the hierarchy of its downward rays itself, beside its
    // darts; the one for the occlusion rays is then started after the placement, on the same thread object.)  The
    // worker's status and message come back through `mesh`.
    struct Mesh {
        std::mutex mu;
        std::condition_variable cv;
        int stage = 0;                                        // 1: obj / path_model / tri are ready; 2: bvh too (or failed)
        int rc = C2B_OK;
        std::string err;
        c2b_obj *obj = nullptr;
        int64_t path_model = -1, n_tri = 0;
        std::vector<float> tri;
        c2b_bvh *bvh = nullptr;
        bool want_bvh = false;
        void fail_with(const std::string &m) { std::lock_guard<std::mutex> lk(mu); rc = -1; err = m; stage = 2; cv.notify_all(); }
        void reach(int s) { std::lock_guard<std::mutex> lk(mu); stage = s; cv.notify_all(); }
        void wait_for(int s) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stage >= s; }); }
    } mesh;
    auto build_bvh = [&]() {
        if (c2b_bvh_build(mesh.tri.data(), mesh.n_tri, &mesh.bvh) != C2B_OK) mesh.fail_with(c2b_last_error());
        else mesh.reach(2);
    };
    std::thread worker([&]() {
        if (c2b_obj_load(a.positional[0].c_str(), &mesh.obj) != C2B_OK) return mesh.fail_with(c2b_last_error());   // the message is per thread
        if (a.has("path")) {
            const std::string want = a.opt.at("path");
            std::string names;
            for (int64_t m = 0; m < c2b_obj_model_count(mesh.obj); ++m) {
                const std::string name = c2b_obj_model_name(mesh.obj, m);
                if (name == want && mesh.path_model < 0) mesh.path_model = m;
                names += (m ? ", " : "") + name;
            }
            if (mesh.path_model < 0) return mesh.fail_with("Could not find a path named " + want + ". Available model names are " + names);
        }
        if (a.has("move-to-origin") && c2b_obj_move_to_origin(mesh.obj, mesh.path_model) != C2B_OK) return mesh.fail_with(c2b_last_error());
        if (c2b_obj_triangles(mesh.obj, mesh.path_model, nullptr, &mesh.n_tri) != C2B_OK) return mesh.fail_with(c2b_last_error());
        mesh.tri.resize((size_t)mesh.n_tri * 9 + 1);
        if (c2b_obj_triangles(mesh.obj, mesh.path_model, mesh.tri.data(), &mesh.n_tri) != C2B_OK) return mesh.fail_with(c2b_last_error());
        mesh.want_bvh = mesh.n_tri >= 4096;                   // smaller meshes: the all-triangles kernel
        mesh.reach(1);
        if (mesh.want_bvh && mesh.path_model >= 0) build_bvh();
    });
    struct JoinWorker { std::thread &t; Mesh &m; ~JoinWorker() { if (t.joinable()) t.join(); if (m.bvh) c2b_bvh_free(m.bvh); } } join_worker{worker, mesh};
    c2b_problem *p = nullptr;
    const int create_rc = c2b_problem_create((int)a.i("device", 0), &p);
    const std::string create_err = create_rc != C2B_OK ? c2b_last_error() : "";
    mesh.wait_for(1);
    if (mesh.rc != C2B_OK) die(mesh.err);
    if (create_rc != C2B_OK) die(create_err);
    timer.mark("load .obj + triangles || HIP runtime start");
    c2b_obj *obj = mesh.obj;
    const int64_t path_model = mesh.path_model, n_tri = mesh.n_tri;
    const std::vector<float> &tri = mesh.tri;
    const bool own_bvh = mesh.want_bvh;

    int64_t n_cam = 0;
    std::vector<double> pos, dir;
    if (path_model >= 0) {
        n_cam = num_cameras;
        pos.resize((size_t)n_cam * 3 + 1); dir.resize((size_t)n_cam * 9 + 1);
        ck(c2b_generate_cameras_path(obj, path_model, num_cameras, step_size, seed, pos.data(), dir.data()));
    } else {
        // 2 * num_cameras disks at the densest packing bound the sample count (plus a rim for the open boundary)
        int64_t cap = 2 * num_cameras + 8 * (int64_t)std::sqrt((double)(2 * num_cameras)) + 64;
        pos.resize((size_t)cap * 3 + 1); dir.resize((size_t)cap * 9 + 1);
        // (the placement builds its own hierarchy for the downward rays while it throws its darts -- the longer of the
        // two; the one for the occlusion rays is started afterwards and hides behind the sampling and the sweep)
        ck(c2b_generate_cameras_poisson(tri.data(), n_tri, num_cameras, height, ground, seed, cap, pos.data(), dir.data(), &n_cam));
        if (n_cam > cap) {                                  // same seed => same cameras
            cap = n_cam;
            pos.resize((size_t)cap * 3 + 1); dir.resize((size_t)cap * 9 + 1);
            ck(c2b_generate_cameras_poisson(tri.data(), n_tri, num_cameras, height, ground, seed, cap, pos.data(), dir.data(), &n_cam));
        }
        if (own_bvh) {                                      // the worker has finished (stage 1 was its last): reuse the thread object
            worker.join();
            worker = std::thread(build_bvh);
        }
    }
    c2b_obj_free(obj);
    timer.mark("triangles + camera placement");
    std::printf("Generated %lld cameras\n", (long long)n_cam);

    HostProblem hp;
    hp.n_cam = n_cam;
    hp.cams15.resize((size_t)n_cam * 15 + 1);
    if (n_cam) ck(c2b_problem_from_position_direction(p, n_cam, pos.data(), dir.data(), hp.cams15.data()));
    ck(c2b_modify_intrinsics(hp.cams15.data(), n_cam, istart, iend, seed + 1));
    timer.mark("from_position_direction + intrinsics");
    std::printf("Modified intrinsics\n");

    std::vector<uint64_t> rows((size_t)n_cam + 1, 0);
    ck(c2b_problem_upload(p, n_cam, hp.cams15.data(), 0, nullptr, rows.data(), nullptr, nullptr));
    // generate_world_points_uniform on the resident cameras (r04: candidates evaluated on the device, the host sampler's
    // points bit for bit; C2B_HOST_SAMPLER=1 = the host sampler and a second upload)
    if (std::getenv("C2B_HOST_SAMPLER")) {
        std::vector<double> centers((size_t)n_cam * 3 + 1);
        if (n_cam) ck(c2b_problem_centers(p, centers.data()));
        hp.pts.resize((size_t)num_points * 3 + 1);
        ck(c2b_generate_world_points(tri.data(), n_tri, centers.data(), n_cam, num_points, max_dist, seed + 2, hp.pts.data(), &hp.n_pts));
        ck(c2b_problem_upload(p, n_cam, hp.cams15.data(), hp.n_pts, hp.pts.data(), rows.data(), nullptr, nullptr));
    } else {
        ck(c2b_problem_generate_world_points(p, tri.data(), n_tri, num_points, max_dist, seed + 2, &hp.n_pts));
    }
    timer.mark("world points");
    std::printf("Generated %lld world points\n", (long long)hp.n_pts);

    // visibility_graph, src/generate.rs:424-481: the points within max_dist of every camera that pass the predicate (the
    // cell list of the synthetic generators stands in for rstar here too since r04: the same lists as the brute-force
    // sweep of every camera against every point, 4x sooner at these sizes; C2B_DENSE_SWEEP=1 selects the sweep), then
    // the occlusion rays
    hp.row_ptr.assign((size_t)n_cam + 1, 0);
    if (std::getenv("C2B_DENSE_SWEEP")) ck(c2b_problem_visibility_dense(p, max_dist, hp.row_ptr.data()));
    else ck(c2b_problem_visibility_within_distance(p, max_dist, 0, 0.0, 0.0, hp.row_ptr.data()));
    timer.mark("candidates within max_dist + predicate");
    if (own_bvh) {
        mesh.wait_for(2);
        if (mesh.rc != C2B_OK) die(mesh.err);
        ck(c2b_problem_visibility_dense_occlude_bvh(p, mesh.bvh, hp.row_ptr.data()));
    } else {
        ck(c2b_problem_visibility_dense_occlude(p, tri.data(), n_tri, hp.row_ptr.data()));
    }
    timer.mark("occlusion rays + compaction (the hierarchy was built beside the phases above)");
    const size_t n_edges = (size_t)hp.row_ptr[(size_t)n_cam];
    std::printf("Computed visibility graph with %zu edges\n", n_edges);

    // from_visibility and cull stay on the device; one download at the end
    ck(c2b_problem_adopt_visibility(p));
    if (!a.has("no-lcc")) ck(c2b_problem_cull(p, a.has("exact-lcc") ? 0 : 1));
    timer.mark("from_visibility + cull (device)");
    int64_t nc = 0, np = 0, no = 0;
    ck(c2b_problem_sizes(p, &nc, &np, &no));
    if (nc == 0 || np == 0) die("EmptyProblem(\"No cameras remain\")");
    std::printf("Computed LCC with %lld cameras, %lld points, %lld edges\n", (long long)nc, (long long)np, (long long)no);

    double l1 = 0;
    ck(c2b_problem_total_reprojection_error(p, 1.0, &l1));
    std::printf("Total reprojection error: %s\n", display_f64(l1).c_str());
    timer.mark("error");
    // BAProblem::write of the resident problem: to_vec and a .bbal's file image on the device, nothing downloaded
    ck(c2b_problem_write(p, a.positional[1].c_str(), -1));
    timer.mark("write");
    c2b_problem_destroy(p);
    return 0;
}

// run_ply, src/bin/city2ba.rs:441-445
int run_ply(int argc, char **argv) {
    const Args a = parse(argc, argv, 2, {}, {"device"});
    if (a.positional.size() != 2) die("The following required arguments were not provided:\n    <FILE> <OUT>");
    c2b_balfile *f = nullptr;
    ck(c2b_bal_read(a.positional[0].c_str(), &f));
    int64_t n_cam, n_pts, n_obs;
    ck(c2b_bal_sizes(f, &n_cam, &n_pts, &n_obs));
    std::vector<double> bal9((size_t)n_cam * 9 + 1), pts((size_t)n_pts * 3 + 1), uv((size_t)n_obs * 2 + 1);
    std::vector<uint64_t> row_ptr((size_t)n_cam + 1), pt_idx((size_t)n_obs + 1);
    ck(c2b_bal_copy(f, bal9.data(), pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    c2b_bal_close(f);
    c2b_problem *p = nullptr;
    ck(create_problem((int)a.i("device", 0), &p));
    ck(c2b_problem_upload_bal(p, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data()));
    std::vector<double> centers((size_t)n_cam * 3 + 1);
    ck(c2b_problem_centers(p, centers.data()));
    c2b_problem_destroy(p);
    ck(c2b_ply_write(a.positional[1].c_str(), n_cam, centers.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data()));
    return 0;
}

void usage() {
    std::printf("city2ba (MI355X build, %s)\nTools for generating synthetic bundle adjustment problems.\n\n"
                "USAGE:\n    city2ba <SUBCOMMAND>\n\nSUBCOMMANDS:\n"
                "    synthetic         Generate a synthetic bundle adjustment problem from an grid of city blocks.\n"
                "    synthetic-line    Generate a synthetic bundle adjustment problem on a line.\n"
                "    noise             Add noise to a bundle adjustment problem.\n"
                "    generate          Generate a synthetic bundle adjustment problem from a 3D model.\n"
                "    ply               Convert a .bal or .bbal to a .ply for visualization.\n",
                c2b_version());
}

// flags of each subcommand with the reference's defaults (src/bin/city2ba.rs:33-260)
const char *subcommand_help(const std::string &sub) {
    if (sub == "synthetic")
        return "city2ba synthetic <OUTPUT>\n"
               "    --blocks <N>              city blocks per side [5]\n"
               "    --cameras-per-block <N>   [10]        --points-per-block <N>   [10]\n"
               "    --block-length <X>        [20]        --block-inset <X>        [1]\n"
               "    --camera-height <X>       [1]         --point-height <X>       [1]\n"
               "    --max-dist <X>            maximum camera-point distance [10]\n";
    if (sub == "synthetic-line")
        return "city2ba synthetic-line <OUTPUT>\n"
               "    --cameras <N> [10]   --points <N> [10]   --length <X> [20]   --point-offset <X> [1]\n"
               "    --camera-height <X> [1]   --point-height <X> [1]   --max-dist <X> [10]\n";
    if (sub == "noise")
        return "city2ba noise <FILE> <OUT>\n"
               "    --rotation-std <X> --translation-std <X> --point-std <X> --observation-std <X>   Gaussian noise [0]\n"
               "    --drift-strength <X> --drift-angle <X> --drift-std <X> [--fixed-drift]            drift [0]\n"
               "    --sin-strength <X> [0]  --sin-frequency <X> [1]                                  sine displacement\n"
               "    --mismatch-chance <X> [0]  --drop-features <X> [1]  --split-landmarks <X> [0]  --join-landmarks <X> [0]\n"
               "    --seed <N>                every random draw is seeded (default: std::random_device)\n"
               "    --gpus <N> | --devices <a,b,...>   shard the problem over N GPUs of this node (contiguous camera ranges,\n"
               "                              points replicated, statistics and errors through RCCL); the same file for every N up to rounding\n";
    if (sub == "generate")
        return "city2ba generate <FILE.obj> <OUT>\n"
               "    --cameras <N> [100]   --points <N> [1000]   --max-dist <X> [100]\n"
               "    --intrinsics-start <f,k1,k2> [1,0,0]   --intrinsics-end <f,k1,k2> [1,0,0]\n"
               "    --path <NAME> [--step-size <X> [0]]    cameras along the polyline model NAME\n"
               "    --ground <X> [0]  --height <X> [1]     Poisson placement (without --path)\n"
               "    --move-to-origin   --no-lcc   --exact-lcc (extension)   --seed <N>\n";
    if (sub == "ply") return "city2ba ply <FILE> <OUT.ply>\n";
    return nullptr;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2 || !std::strcmp(argv[1], "--help") || !std::strcmp(argv[1], "-h") || !std::strcmp(argv[1], "help")) {
        usage();
        return argc < 2 ? 1 : 0;
    }
    const std::string sub = argv[1];
    c2b_host_set_io_threads(env_int("C2B_IO_THREADS", 0));               // the handle-less host entries (c2b_bal_read / _write)
    for (int k = 2; k < argc; ++k)
        if (!std::strcmp(argv[k], "--help") || !std::strcmp(argv[k], "-h")) {
            const char *h = subcommand_help(sub);
            if (h) { std::printf("%s    --device <N>              GPU index [0]; C2B_TIMING=1 prints phase times\n", h); return 0; }
        }
    if (sub == "synthetic") return run_synthetic(argc, argv);
    if (sub == "synthetic-line") return run_synthetic_line(argc, argv);
    if (sub == "noise") return run_noise(argc, argv);
    if (sub == "generate") return run_generate(argc, argv);
    if (sub == "ply") return run_ply(argc, argv);
    die("The subcommand '" + sub + "' wasn't recognized");
}
