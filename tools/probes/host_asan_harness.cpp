// The host-only C++ of this round under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool offers no device
// sanitizer; these headers need no HIP): decimal.hpp (format / parse, tables), the threaded text writer / reader of
// host_baproblem.hpp, the .obj loader, the batched Poisson darts with their threaded pre-test and ray casts, the
// world-point sampler of host_generate.hpp.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread tools/probes/host_asan_harness.cpp -o /tmp/asan_h && /tmp/asan_h
#include "../../city2ba_amd/csrc/host_baproblem.hpp"
#include "../../city2ba_amd/csrc/host_generate.hpp"
#include <cstdio>
#include <random>
using namespace c2b_host;
int main() {
    std::mt19937_64 rng(7);
    // decimal: format -> parse is the identity on random bit patterns
    const auto &T = c2b_dec::host_tables();
    const auto &P = c2b_dec::host_parse_tables();
    long bad = 0;
    for (int i = 0; i < 400000; ++i) {
        uint64_t b = rng(); double v; std::memcpy(&v, &b, 8);
        if (!(v == v) || v - v != 0) continue;
        const c2b_dec::Text t = c2b_dec::describe(v, &T);
        std::string s(t.len, '?'); c2b_dec::emit(t, &s[0]);
        int st = 0; const double w = c2b_dec::parse_f64(s.data(), (int32_t)s.size(), &P, &st);
        uint64_t c; std::memcpy(&c, &w, 8);
        bad += (st != 0 || c != b);
    }
    std::printf("decimal round trips: %ld bad\n", bad);
    // a problem, written and read as text on several threads and as binary
    Graph g; g.stride = 9; g.n_cam = 3000; g.n_pts = 9000;
    std::uniform_real_distribution<double> U(-100, 100);
    g.cams.resize((size_t)g.n_cam * 9); for (auto &x : g.cams) x = U(rng);
    g.pts.resize((size_t)g.n_pts * 3); for (auto &x : g.pts) x = U(rng);
    g.row_ptr.assign(1, 0);
    for (int64_t c = 0; c < g.n_cam; ++c) {
        const int k = (int)(rng() % 120);
        for (int j = 0; j < k; ++j) { g.pt_idx.push_back(rng() % (uint64_t)g.n_pts); g.uv.push_back(U(rng)); g.uv.push_back(U(rng) * 1e-7); }
        g.row_ptr.push_back(g.pt_idx.size());
    }
    std::string err;
    for (int threads : {1, 3, 8}) {
        if (!write_text("/tmp/asan_h.bal", g, &err, threads)) { std::printf("write_text: %s\n", err.c_str()); return 1; }
        Graph h;
        if (!read_text("/tmp/asan_h.bal", h, &err, threads)) { std::printf("read_text: %s\n", err.c_str()); return 1; }
        std::printf("text, %d threads: %s\n", threads, (h.cams == g.cams && h.pts == g.pts && h.pt_idx == g.pt_idx && h.uv == g.uv && h.row_ptr == g.row_ptr) ? "same" : "DIFFERENT");
    }
    if (!write_binary("/tmp/asan_h.bbal", g, &err)) return 1;
    { Graph h; if (!read_binary("/tmp/asan_h.bbal", h, &err)) return 1; std::printf("binary: %s\n", h.uv == g.uv ? "same" : "DIFFERENT"); }
    // a terrain mesh as .obj: the loader's vertex table; Poisson placement (batched darts, threaded rays); world points
    const int n = 60;
    { FILE *f = std::fopen("/tmp/asan_h.obj", "w");
      std::fprintf(f, "o ground\n");
      for (int i = 0; i <= n; ++i) for (int j = 0; j <= n; ++j) std::fprintf(f, "v %d %.3f %d\n", i, 0.2 * ((i * 7 + j * 3) % 5), j);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { const int a = i * (n + 1) + j + 1; std::fprintf(f, "f %d %d %d\nf %d %d %d\n", a, a + n + 1, a + 1, a + n + 1, a + n + 2, a + 1); }
      std::fprintf(f, "o path\nv 1 1 1\nv 50 1 50\nl -2 -1\no tex\nv 0 0 0\nv 1 0 0\nv 0 0 1\nvt 0 0\nvn 0 1 0\nf -3/1/1 -2/1/1 -1/1/1\n");
      std::fclose(f); }
    std::vector<ObjModel> models;
    if (!load_obj("/tmp/asan_h.obj", models, &err)) { std::printf("load_obj: %s\n", err.c_str()); return 1; }
    std::printf("obj: %zu models, %zu + %zu + %zu indices\n", models.size(), models[0].indices.size(), models[1].indices.size(), models[2].indices.size());
    std::vector<float> tri;
    for (size_t k = 0; k < models[0].indices.size(); ++k) for (int c = 0; c < 3; ++c) tri.push_back(models[0].positions[3 * models[0].indices[k] + (size_t)c]);
    CameraSamples cs;
    cameras_poisson(tri, 30000, 1.7, 1000.0, 5, cs);
    std::printf("poisson: %zu cameras\n", cs.size());
    std::vector<double> pts;
    if (!world_points_uniform(tri, cs.pos.data(), (int64_t)cs.size(), 20000, 3.0, 9, pts, &err)) { std::printf("world points: %s\n", err.c_str()); return 1; }
    std::printf("world points: %zu\n", pts.size() / 3);
    std::remove("/tmp/asan_h.bal"); std::remove("/tmp/asan_h.bbal"); std::remove("/tmp/asan_h.obj");
    return 0;
}
