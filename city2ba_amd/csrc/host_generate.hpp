// host_generate.hpp -- host-side (CPU, C++) pieces of the mesh generator around the device sweep
// (SURVEY section 8f row 4): Wavefront .obj loading with tobj 0.1.12's conventions, the camera samplers
// (src/generate.rs:109-280), area-weighted point sampling (:282-420), modify_intrinsics (:530-544),
// move_to_origin (:484-527) and a brute-force ray caster that stands in for Embree's scene.intersect /
// scene.bounds (Embree is not available; a BVH is a separate project -- scenes of the size of
// test_scene.obj have a few hundred triangles).
//
// Every sampler in the reference draws from an unseeded rand::thread_rng(); here a seed is explicit
// (std::mt19937_64).  Only distributions and the deterministic geometry around the draws can match.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <tuple>
#include <vector>

#include "host_bvh.hpp"

namespace c2b_host {

// ---- .obj (tobj 0.1.12 conventions) ---------------------------------------------------------------
// One model per `o` / `g` statement that owns faces or lines; positions are f32 and re-indexed per model by
// unique (v, vt, vn) triple in order of first use; polygons are fan-triangulated; `l a b c` polylines become
// index pairs (a,b), (b,c) -- what generate_cameras_path reads with `.tuples()` (src/generate.rs:121-127).
struct ObjModel {
    std::string name;
    std::vector<float> positions;     // [n][3]
    std::vector<uint32_t> indices;    // triangles (3 per face) or line segments (2 per segment)
    bool lines = false;
};

inline bool load_obj(const char *path, std::vector<ObjModel> &models, std::string *err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { *err = std::string("Could not open file \"") + path + "\""; return false; }
    std::vector<float> v;
    int64_t n_vt = 0, n_vn = 0;
    ObjModel cur;
    cur.name = "unnamed_object";
    std::map<std::tuple<int64_t, int64_t, int64_t>, uint32_t> remap;
    // position-only corners (no texture / normal index: every mesh of the tests and of tools/make_city_obj.py) are
    // looked up in a table by vertex instead of in the map: `seen_in[iv]` = the model that last used vertex iv
    // (a 0.8 M-triangle city spent most of its 218 ms load in the map)
    std::vector<uint32_t> plain_id, seen_in;
    uint32_t model_no = 1;
    auto flush = [&]() {
        if (!cur.indices.empty()) models.push_back(std::move(cur));
        cur = ObjModel();
        remap.clear();
        ++model_no;
    };
    auto vertex = [&](int64_t iv, int64_t it, int64_t in) -> uint32_t {
        if (it < 0 && in < 0) {
            if ((size_t)iv >= seen_in.size()) { seen_in.resize(v.size() / 3, 0u); plain_id.resize(v.size() / 3, 0u); }
            if (seen_in[(size_t)iv] == model_no) return plain_id[(size_t)iv];
            const uint32_t id = (uint32_t)(cur.positions.size() / 3);
            cur.positions.insert(cur.positions.end(), {v[3 * iv], v[3 * iv + 1], v[3 * iv + 2]});
            seen_in[(size_t)iv] = model_no;
            plain_id[(size_t)iv] = id;
            return id;
        }
        auto key = std::make_tuple(iv, it, in);
        auto hit = remap.find(key);
        if (hit != remap.end()) return hit->second;
        const uint32_t id = (uint32_t)(cur.positions.size() / 3);
        cur.positions.insert(cur.positions.end(), {v[3 * iv], v[3 * iv + 1], v[3 * iv + 2]});
        remap[key] = id;
        return id;
    };
    char line[4096];
    bool ok = true;
    while (ok && std::fgets(line, sizeof line, f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *q = p + 1;
            for (int k = 0; k < 3; ++k) v.push_back(std::strtof(q, &q));
        } else if (p[0] == 'v' && p[1] == 't') {
            ++n_vt;
        } else if (p[0] == 'v' && p[1] == 'n') {
            ++n_vn;
        } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t')) {
            flush();
            char *q = p + 1;
            while (*q == ' ' || *q == '\t') ++q;
            size_t len = std::strlen(q);
            while (len && (q[len - 1] == '\n' || q[len - 1] == '\r' || q[len - 1] == ' ')) --len;
            cur.name.assign(q, len);
            if (cur.name.empty()) cur.name = "unnamed_object";
        } else if ((p[0] == 'f' || p[0] == 'l') && (p[1] == ' ' || p[1] == '\t')) {
            const bool is_line = p[0] == 'l';
            std::vector<uint32_t> ids;
            char *q = p + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '\0' || *q == '\n' || *q == '\r') break;
                int64_t idx[3] = {0, 0, 0};
                for (int k = 0; k < 3; ++k) {
                    char *e = q;
                    const long long val = std::strtoll(q, &e, 10);
                    if (e != q) idx[k] = val;
                    q = e;
                    if (*q == '/') ++q; else break;
                }
                const int64_t nv = (int64_t)(v.size() / 3);
                int64_t iv = idx[0] > 0 ? idx[0] - 1 : nv + idx[0];
                const int64_t it = idx[1] > 0 ? idx[1] - 1 : (idx[1] < 0 ? n_vt + idx[1] : -1);
                const int64_t in = idx[2] > 0 ? idx[2] - 1 : (idx[2] < 0 ? n_vn + idx[2] : -1);
                if (idx[0] == 0 || iv < 0 || iv >= nv) { *err = "Load error: face or line index out of range"; ok = false; break; }
                ids.push_back(vertex(iv, it, in));
            }
            if (!ok) break;
            if (is_line) {
                for (size_t k = 0; k + 1 < ids.size(); ++k) { cur.indices.push_back(ids[k]); cur.indices.push_back(ids[k + 1]); }
                if (ids.size() >= 2) cur.lines = true;
            } else {
                for (size_t k = 1; k + 1 < ids.size(); ++k) {      // triangle fan
                    cur.indices.push_back(ids[0]); cur.indices.push_back(ids[k]); cur.indices.push_back(ids[k + 1]);
                }
            }
        }
    }
    std::fclose(f);
    if (!ok) return false;
    flush();
    return true;
}

// triangles of a set of models as packed f32 [n_tri][9]
inline void triangles_of(const std::vector<ObjModel> &models, std::vector<float> &tri9) {
    tri9.clear();
    for (const ObjModel &m : models) {
        if (m.lines) continue;
        for (size_t t = 0; t + 2 < m.indices.size(); t += 3)
            for (int k = 0; k < 3; ++k) {
                const uint32_t i = m.indices[t + k];
                tri9.insert(tri9.end(), {m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]});
            }
    }
}

// move_to_origin, src/generate.rs:484-527: subtract the component-wise minimum over all models (f32)
// `skip` (or -1): a model left out of both the minimum and the move -- run_generate (src/bin/city2ba.rs:493-513)
// clones the --path model out of the list before move_to_origin, so the path keeps its file coordinates
inline void move_to_origin(std::vector<ObjModel> &models, int64_t skip = -1) {
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    for (size_t j = 0; j < models.size(); ++j) {
        if ((int64_t)j == skip) continue;
        const ObjModel &m = models[j];
        for (size_t i = 0; i + 2 < m.positions.size(); i += 3)
            for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], m.positions[i + k]);
    }
    for (size_t j = 0; j < models.size(); ++j) {
        if ((int64_t)j == skip) continue;
        ObjModel &m = models[j];
        for (size_t i = 0; i + 2 < m.positions.size(); i += 3)
            for (int k = 0; k < 3; ++k) m.positions[i + k] = m.positions[i + k] - mn[k];
    }
}

// ---- brute-force stand-in for Embree (f32 rays like embree_rs::Ray) ------------------------------------
// Moeller-Trumbore, no back-face culling; returns the hit distance or a negative value
inline float ray_triangle(const float o[3], const float d[3], const float *t9) {
    const float e1[3] = {t9[3] - t9[0], t9[4] - t9[1], t9[5] - t9[2]};
    const float e2[3] = {t9[6] - t9[0], t9[7] - t9[1], t9[8] - t9[2]};
    const float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (det == 0.0f) return -1.0f;
    const float inv = 1.0f / det;
    const float tv[3] = {o[0] - t9[0], o[1] - t9[1], o[2] - t9[2]};
    const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float w = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
    if (w < 0.0f || u + w > 1.0f) return -1.0f;
    return (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
}

// scene.intersect: nearest hit with t >= 0 (tnear = 0, tfar = inf); returns false when nothing is hit
inline bool cast_ray(const std::vector<float> &tri9, const float o[3], const float d[3], float *t_hit) {
    float best = INFINITY;
    for (size_t t = 0; t + 8 < tri9.size(); t += 9) {
        const float th = ray_triangle(o, d, &tri9[t]);
        if (th >= 0.0f && th < best) best = th;
    }
    *t_hit = best;
    return best < INFINITY;
}

inline void scene_bounds(const std::vector<float> &tri9, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (size_t i = 0; i + 2 < tri9.size(); i += 3)
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], tri9[i + k]); hi[k] = std::max(hi[k], tri9[i + k]); }
}

// ---- cgmath pieces the samplers need ---------------------------------------------------------------------
// Basis3::between_vectors(a, b) = Quaternion::from_arc(a, b, None) -> matrix (column-major)
inline void basis_between_vectors(const double a[3], const double b[3], double m[9]) {
    const double mag_avg = std::sqrt(((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) * ((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]));
    const double dot = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    double s, x, y, z;
    auto ulps_eq = [](double p, double q) { return std::fabs(p - q) <= 4 * 2.220446049250313e-16 * std::max(std::fabs(p), std::fabs(q)) || std::fabs(p - q) <= 2.220446049250313e-16; };
    if (ulps_eq(dot, mag_avg)) {
        s = 1; x = y = z = 0;
    } else if (ulps_eq(dot, -mag_avg)) {
        // 180 degrees about an axis orthogonal to a
        double ax[3] = {0.0, -a[2], a[1]};                           // unit_x x a
        if (ax[0] == 0.0 && std::fabs(ax[1]) < 1e-300 && std::fabs(ax[2]) < 1e-300) { ax[0] = a[2]; ax[1] = 0.0; ax[2] = -a[0]; }   // unit_y x a
        const double inv = 1.0 / std::sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2]);
        const double h = 3.14159265358979323846 * 0.5;                 // Rad::turn_div_2() / 2
        s = std::cos(h); x = ax[0] * inv * std::sin(h); y = ax[1] * inv * std::sin(h); z = ax[2] * inv * std::sin(h);
    } else {
        s = mag_avg + dot;
        x = a[1] * b[2] - a[2] * b[1]; y = a[2] * b[0] - a[0] * b[2]; z = a[0] * b[1] - a[1] * b[0];
        const double inv = 1.0 / std::sqrt(s * s + ((x * x + y * y) + z * z));
        s *= inv; x *= inv; y *= inv; z *= inv;
    }
    const double x2 = x + x, y2 = y + y, z2 = z + z;
    const double xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const double sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    const double t[9] = {1.0 - yy2 - zz2, xy2 + sz2, xz2 - sy2, xy2 - sz2, 1.0 - xx2 - zz2, yz2 + sx2,
                         xz2 + sy2, yz2 - sx2, 1.0 - xx2 - yy2};
    std::copy(t, t + 9, m);
}

inline double uniform01(std::mt19937_64 &rng) { return std::uniform_real_distribution<double>(0.0, 1.0)(rng); }

// rand's WeightedIndex: index i with probability w[i] / sum(w)
struct WeightedIndex {
    std::vector<double> cum;
    explicit WeightedIndex(const std::vector<double> &w) {
        double acc = 0;
        for (double x : w) { acc += x; cum.push_back(acc); }
    }
    size_t at(double u01) const {
        const double u = u01 * cum.back();
        size_t i = (size_t)(std::upper_bound(cum.begin(), cum.end(), u) - cum.begin());
        return std::min(i, cum.size() - 1);
    }
    size_t sample(std::mt19937_64 &rng) const { return at(uniform01(rng)); }
};

// Counter-based draws for loops that run on several threads: candidate k of a seeded loop owns its own splitmix64
// stream, so the outcome does not depend on how candidates are spread over threads.
struct CounterRng {
    uint64_t s;
    CounterRng(uint64_t seed, uint64_t k) : s(seed * 0xD6E8FEB86659FD93ull + k * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform01() { return (double)(next() >> 11) * 0x1.0p-53; }
};

struct CameraSamples {
    std::vector<double> pos, dir;      // [n][3], [n][9] column-major
    size_t size() const { return pos.size() / 3; }
};

// path segments of a line model in f64 (positions are f32 in the file)
inline void path_segments(const ObjModel &path, std::vector<double> &seg) {
    seg.clear();
    for (size_t k = 0; k + 1 < path.indices.size(); k += 2)
        for (int e = 0; e < 2; ++e) {
            const uint32_t i = path.indices[k + e];
            for (int c = 0; c < 3; ++c) seg.push_back((double)path.positions[3 * i + c]);
        }
}

inline void push_path_camera(CameraSamples &out, const double x[3], const double dir[3], double along) {
    const double len = std::sqrt((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    const double dn[3] = {dir[0] * (1.0 / len), dir[1] * (1.0 / len), dir[2] * (1.0 / len)};
    const double fwd[3] = {0.0, 0.0, -1.0};
    double m[9];
    basis_between_vectors(dn, fwd, m);
    for (int c = 0; c < 3; ++c) out.pos.push_back(x[c] + along * dir[c]);
    out.dir.insert(out.dir.end(), m, m + 9);
}

// generate_cameras_path, src/generate.rs:109-148
inline bool cameras_path(const ObjModel &path, int64_t num_cameras, uint64_t seed, CameraSamples &out, std::string *err) {
    std::vector<double> seg;
    path_segments(path, seg);
    if (seg.empty()) { *err = "path model has no line segments"; return false; }
    std::vector<double> len;
    for (size_t s = 0; s + 5 < seg.size(); s += 6) {
        const double d[3] = {seg[s + 3] - seg[s], seg[s + 4] - seg[s + 1], seg[s + 5] - seg[s + 2]};
        len.push_back(std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]));
    }
    std::mt19937_64 rng(seed);
    const WeightedIndex dist(len);
    for (int64_t k = 0; k < num_cameras; ++k) {
        const size_t i = dist.sample(rng);
        const double d = uniform01(rng);
        const double *x = &seg[6 * i];
        const double dir[3] = {x[3] - x[0], x[4] - x[1], x[5] - x[2]};
        push_path_camera(out, x, dir, d);
    }
    return true;
}

// generate_cameras_path_step, src/generate.rs:152-213
inline bool cameras_path_step(const ObjModel &path, int64_t num_cameras, double step_size, CameraSamples &out,
                              std::string *err, double *total_length) {
    std::vector<double> seg;
    path_segments(path, seg);
    if (seg.empty()) { *err = "path model has no line segments"; return false; }
    auto seglen = [&](size_t i) {
        const double d[3] = {seg[6 * i + 3] - seg[6 * i], seg[6 * i + 4] - seg[6 * i + 1], seg[6 * i + 5] - seg[6 * i + 2]};
        return std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    };
    const size_t n_seg = seg.size() / 6;
    double total = 0;
    for (size_t i = 0; i < n_seg; ++i) total += seglen(i);
    *total_length = total;
    if (!((double)num_cameras * step_size <= total)) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "Length of path %g is less than the number of cameras (%lld) times the step size (%g) %g",
                      total, (long long)num_cameras, step_size, (double)num_cameras * step_size);
        *err = buf;
        return false;
    }
    size_t si = 0;
    double dist = 0.0;
    for (int64_t k = 0; k < num_cameras; ++k) {
        const double *x = &seg[6 * si];
        double dir[3] = {x[3] - x[0], x[4] - x[1], x[5] - x[2]};
        double mag = seglen(si);
        push_path_camera(out, x, dir, dist / mag);
        dist += step_size;
        while (dist >= mag) {
            ++si;
            dist -= mag;
            if (si >= n_seg) {           // the reference would index out of bounds (panic) on the last camera
                si = n_seg - 1;
                dist = std::min(dist, seglen(si));
                break;
            }
            mag = seglen(si);
        }
    }
    return true;
}

// Poisson-disk samples in the unit square by dart throwing on a background grid (the reference uses the
// `poisson` crate's Ebeida algorithm with_samples(n, 1.0, Normal); same kind of blue-noise set, different draws)
inline void poisson_unit_square(int64_t n_target, std::mt19937_64 &rng, std::vector<double> &xy) {
    xy.clear();
    if (n_target <= 0) return;
    // The poisson crate's with_samples(n, 1.0, ..) picks the minimum distance r at which n disks of radius r/2 would
    // tile the unit square at the densest (hexagonal) packing, pi / sqrt(12): r^2 = 2 / (sqrt(3) n) = 1.1547 / n.
    // A maximal random set at that distance reaches the jamming density 0.547, i.e. ~0.6 n samples -- which is why
    // the reference asks for 2x the cameras it wants (src/generate.rs:231).
    const double r = std::sqrt(1.1547005383792515 / (double)n_target);
    const double cell = r / std::sqrt(2.0);
    const int64_t g = std::max<int64_t>(1, (int64_t)std::ceil(1.0 / cell));
    std::vector<int64_t> grid((size_t)(g * g), -1);
    const int64_t attempts = 30 * n_target + 1000;
    // A dart is rejected by ANY sample closer than r, so the 5 x 5 cells around it may be looked at in any order: nearest
    // cells first (its own, then the ring around it) -- once the square is nearly full, 19 of 20 darts die at the first
    // or second cell instead of after 25 (the same samples as the row-by-row scan: 6 M darts for 100 000 cameras).
    int nb[25][2];
    {
        int k = 0;
        for (int d2 = 0; d2 <= 8; ++d2)
            for (int ox = -2; ox <= 2; ++ox)
                for (int oy = -2; oy <= 2; ++oy)
                    if (ox * ox + oy * oy == d2) { nb[k][0] = ox; nb[k][1] = oy; ++k; }
    }
    // A dart that an EXISTING sample rejects stays rejected whatever happens later (samples are only ever added), and
    // drawing a dart does not depend on the fate of the one before it.  So the darts go in batches: all host threads
    // draw a batch's coordinates and test them against the samples present at its start, and only the survivors -- 1 in
    // 20 once the square fills up -- go through the sequential test-and-insert, in dart order.  The same samples as one
    // dart at a time, whatever the thread count.
    auto rejected_by = [&](double x, double y, int64_t gx, int64_t gy) {
        for (int k = 0; k < 25; ++k) {
            const int64_t ix = gx + nb[k][0], iy = gy + nb[k][1];
            if (ix < 0 || iy < 0 || ix >= g || iy >= g) continue;
            const int64_t j = grid[(size_t)(ix * g + iy)];
            if (j >= 0) {
                const double dx = xy[2 * j] - x, dy = xy[2 * j + 1] - y;
                if (dx * dx + dy * dy < r * r) return true;
            }
        }
        return false;
    };
    // Dart k draws its coordinates from its own counter-based stream (CounterRng: the generator that the world-point
    // sampler uses), keyed by one draw of the caller's generator: 12 M sequential Mersenne-twister draws were most of
    // what was left of this function (r04), and a dart's coordinates never depended on the darts before it.
    const uint64_t dart_seed = rng();
    const int64_t batch = 1 << 18;                            // (a thread per share and batch: larger batches, fewer thread starts)
    const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<double> bx((size_t)batch), by((size_t)batch);
    std::vector<char> dead((size_t)batch);
    for (int64_t a0 = 0; a0 < attempts; a0 += batch) {
        const int64_t nb_ = std::min(batch, attempts - a0);
        // the pre-test pays once most darts die (before that nearly every dart survives it)
        const bool pre = (int64_t)(xy.size() / 2) * 4 > n_target;
        auto draw_and_test = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                CounterRng r(dart_seed, (uint64_t)(a0 + i));
                const double x = r.uniform01(), y = r.uniform01();
                bx[(size_t)i] = x; by[(size_t)i] = y;
                dead[(size_t)i] = (pre && rejected_by(x, y, std::min(g - 1, (int64_t)(x / cell)), std::min(g - 1, (int64_t)(y / cell)))) ? 1 : 0;
            }
        };
        if (T > 1) {
            run_threads(T, [&](int t) { draw_and_test(nb_ * t / T, nb_ * (t + 1) / T); });
        } else {
            draw_and_test(0, nb_);
        }
        for (int64_t i = 0; i < nb_; ++i) {
            if (pre && dead[(size_t)i]) continue;
            const double x = bx[(size_t)i], y = by[(size_t)i];
            const int64_t gx = std::min(g - 1, (int64_t)(x / cell)), gy = std::min(g - 1, (int64_t)(y / cell));
            if (rejected_by(x, y, gx, gy)) continue;
            grid[(size_t)(gx * g + gy)] = (int64_t)(xy.size() / 2);
            xy.push_back(x); xy.push_back(y);
        }
    }
}

// generate_cameras_poisson, src/generate.rs:217-280 (incl. its `pt[2] < lower_y + ground` test at :264)
// `ready`: a hierarchy over the same triangles the caller already holds (the command line builds one for the occlusion
// rays anyway), or nullptr: built here, on a second thread, while this one throws the darts.  The downward rays are
// cast on all host threads -- each sample's ray is independent and the hits are gathered in sample order, so the
// cameras do not depend on the thread count (0.56 -> see DESIGN 8 for 100 000 cameras on 829 k triangles).
inline void cameras_poisson(const std::vector<float> &tri9, int64_t num_points, double height, double ground,
                            uint64_t seed, CameraSamples &out, const Bvh *ready = nullptr) {
    std::mt19937_64 rng(seed);
    // meshes beyond a few dozen triangles are searched through the hierarchy (same nearest hit, see host_bvh.hpp)
    const bool use_bvh = ready || tri9.size() / 9 >= 64;
    Bvh built;
    std::thread builder;
    if (use_bvh && !ready) builder = std::thread([&]() { bvh_build(tri9.data(), (int64_t)(tri9.size() / 9), built); });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{builder};
    std::vector<double> samples;
    poisson_unit_square(num_points * 2, rng, samples);
    float lo[3], hi[3];
    scene_bounds(tri9, lo, hi);
    const double start[3] = {(double)hi[0], (double)hi[1] + 0.1, (double)hi[2]};
    const double delta[3] = {(double)(hi[0] - lo[0]), 0.0, (double)(hi[2] - lo[2])};
    if (builder.joinable()) builder.join();
    const Bvh &bvh = ready ? *ready : built;
    const int64_t n_s = (int64_t)(samples.size() / 2);
    std::vector<double> hit_pos((size_t)n_s * 3);
    std::vector<char> hit((size_t)n_s, 0);
    auto cast_range = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const size_t s = (size_t)(2 * i);
            const double origin[3] = {start[0] - delta[0] * samples[s], start[1] - delta[1] * 0.0, start[2] - delta[2] * samples[s + 1]};
            const float of[3] = {(float)origin[0], (float)origin[1], (float)origin[2]};
            const float df[3] = {0.0f, -1.0f, 0.0f};
            float t;
            if (use_bvh ? bvh_cast_ray(bvh, of, df, &t) : cast_ray(tri9, of, df, &t)) {
                const double pt[3] = {origin[0] + 0.0 * (double)t + 0.0, origin[1] + -1.0 * (double)t + height, origin[2] + 0.0 * (double)t + 0.0};
                if (pt[2] < (double)lo[1] + ground) { hit[(size_t)i] = 1; std::copy(pt, pt + 3, &hit_pos[3 * (size_t)i]); }
            }
        }
    };
    const int T = (int)std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), std::max<int64_t>(1, n_s / 2048));
    if (T <= 1) cast_range(0, n_s);
    else run_threads(T, [&](int t) { cast_range(n_s * t / T, n_s * (t + 1) / T); });
    std::vector<double> positions;
    for (int64_t i = 0; i < n_s; ++i)
        if (hit[(size_t)i]) positions.insert(positions.end(), &hit_pos[3 * (size_t)i], &hit_pos[3 * (size_t)i] + 3);
    for (size_t p = 0; p + 2 < positions.size(); p += 3) {
        const double a = uniform01(rng) * (2.0 * 3.14159265358979323846);
        const double s = std::sin(a), c = std::cos(a);
        const double m[9] = {c, 0.0, -s, 0.0, 1.0, 0.0, s, 0.0, c};     // Basis3::from_angle_y
        out.pos.insert(out.pos.end(), &positions[p], &positions[p] + 3);
        out.dir.insert(out.dir.end(), m, m + 9);
    }
}

// modify_intrinsics, src/generate.rs:530-544: intrin = start + v (.) (end - start), v ~ U[0,1)^3
inline void modify_intrinsics(double *cams15, int64_t n, const double start[3], const double end[3], uint64_t seed) {
    std::mt19937_64 rng(seed);
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double v = uniform01(rng);
            cams15[15 * i + 12 + k] = start[k] + v * (end[k] - start[k]);
        }
}

// generate_world_points_uniform, src/generate.rs:356-420
inline bool world_points_uniform(const std::vector<float> &tri9, const double *centers, int64_t n_cam, int64_t num_points,
                                 double max_dist, uint64_t seed, std::vector<double> &pts, std::string *err) {
    pts.clear();
    if (n_cam == 0) {
        *err = "Cannot generate world points with 0 cameras. Try increasing the number of cameras generated (via --cameras).";
        return false;
    }
    const size_t n_tri = tri9.size() / 9;
    if (n_tri == 0) { *err = "the model has no triangles"; return false; }
    std::vector<double> areas(n_tri);
    for (size_t t = 0; t < n_tri; ++t) {
        const float *q = &tri9[9 * t];
        const double a[3] = {(double)q[3] - (double)q[0], (double)q[4] - (double)q[1], (double)q[5] - (double)q[2]};
        const double b[3] = {(double)q[6] - (double)q[0], (double)q[7] - (double)q[1], (double)q[8] - (double)q[2]};
        const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        areas[t] = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) / 2.0;
    }
    // cameras binned on a grid of cell max_dist for the "within max_dist of any camera" test (rstar in the reference)
    // (an open-addressing table from cell coordinates to a run of the cell-sorted camera list)
    const double cs = max_dist > 0 ? max_dist : 1.0;
    struct Cell { int64_t x, y, z; };
    auto cell = [&](const double *p) { return Cell{(int64_t)std::floor(p[0] / cs), (int64_t)std::floor(p[1] / cs), (int64_t)std::floor(p[2] / cs)}; };
    auto hash = [](int64_t x, int64_t y, int64_t z) {
        uint64_t h = (uint64_t)x * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)y + 0x7F4A7C159E3779B9ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((uint64_t)z + 0x165667B19E3779F9ull) * 0xFF51AFD7ED558CCDull;
        return h ^ (h >> 29);
    };
    std::vector<Cell> ccell((size_t)n_cam);
    std::vector<int64_t> order((size_t)n_cam);
    for (int64_t c = 0; c < n_cam; ++c) { ccell[(size_t)c] = cell(centers + 3 * c); order[(size_t)c] = c; }
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        const Cell &p = ccell[(size_t)a], &q = ccell[(size_t)b];
        return std::tie(p.x, p.y, p.z, a) < std::tie(q.x, q.y, q.z, b);
    });
    struct Slot { Cell c; int64_t begin, end; };
    size_t cap = 16;
    while (cap < 2 * (size_t)n_cam) cap <<= 1;
    std::vector<Slot> table(cap, Slot{{0, 0, 0}, 0, -1});                      // end < 0: empty
    for (int64_t i = 0; i < n_cam;) {
        const Cell k = ccell[(size_t)order[(size_t)i]];
        int64_t j = i;
        while (j < n_cam && ccell[(size_t)order[(size_t)j]].x == k.x && ccell[(size_t)order[(size_t)j]].y == k.y &&
               ccell[(size_t)order[(size_t)j]].z == k.z)
            ++j;
        size_t s = (size_t)hash(k.x, k.y, k.z) & (cap - 1);
        while (table[s].end >= 0) s = (s + 1) & (cap - 1);
        table[s] = Slot{k, i, j};
        i = j;
    }
    auto near_camera = [&](const double *p) {
        const Cell k = cell(p);
        for (int64_t dx = -1; dx <= 1; ++dx)
            for (int64_t dy = -1; dy <= 1; ++dy)
                for (int64_t dz = -1; dz <= 1; ++dz) {
                    const int64_t x = k.x + dx, y = k.y + dy, z = k.z + dz;
                    size_t s = (size_t)hash(x, y, z) & (cap - 1);
                    while (table[s].end >= 0 && !(table[s].c.x == x && table[s].c.y == y && table[s].c.z == z)) s = (s + 1) & (cap - 1);
                    if (table[s].end < 0) continue;
                    for (int64_t i = table[s].begin; i < table[s].end; ++i) {
                        const int64_t c = order[(size_t)i];
                        const double e[3] = {centers[3 * c] - p[0], centers[3 * c + 1] - p[1], centers[3 * c + 2] - p[2]};
                        if ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2] <= max_dist * max_dist) return true;
                    }
                }
        return false;
    };
    // The reference's loop (draw a triangle by area, a point in it, keep it if a camera is near, stop at num_points
    // successes or 10 * num_points failures) with candidate k drawing from its own counter-based stream: candidates
    // are evaluated a chunk at a time on all threads, then accepted in candidate order exactly as a sequential
    // loop would.
    const WeightedIndex dist(areas);
    int64_t fail = 0;
    const int64_t fail_threshold = 10 * num_points;
    const unsigned n_threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int64_t chunk = std::max<int64_t>(4096, std::min<int64_t>(num_points, 1 << 20));
    std::vector<double> cand((size_t)chunk * 3);
    std::vector<uint8_t> ok((size_t)chunk);
    for (int64_t k0 = 0; (int64_t)(pts.size() / 3) < num_points && fail < fail_threshold; k0 += chunk) {
        auto work = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                CounterRng rng(seed, (uint64_t)(k0 + i));
                const size_t t = dist.at(rng.uniform01());
                const float *q = &tri9[9 * t];
                double rx = rng.uniform01(), ry = rng.uniform01();
                if (rx + ry > 1.0) { rx = 1.0 - rx; ry = 1.0 - ry; }
                double *p = &cand[3 * (size_t)i];
                for (int c = 0; c < 3; ++c)
                    p[c] = (double)q[c] + rx * ((double)q[3 + c] - (double)q[c]) + ry * ((double)q[6 + c] - (double)q[c]);
                ok[(size_t)i] = near_camera(p) ? 1 : 0;
            }
        };
        const int64_t per = (chunk + n_threads - 1) / n_threads;
        run_threads((int)std::min<int64_t>(n_threads, (chunk + per - 1) / std::max<int64_t>(per, 1)),
                    [&](int t) { work((int64_t)t * per, std::min(chunk, (int64_t)(t + 1) * per)); });
        for (int64_t i = 0; i < chunk && (int64_t)(pts.size() / 3) < num_points && fail < fail_threshold; ++i) {
            if (ok[(size_t)i]) pts.insert(pts.end(), &cand[3 * (size_t)i], &cand[3 * (size_t)i] + 3);
            else ++fail;
        }
    }
    if (fail >= fail_threshold && num_points > 0) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "Failed to generate enough points. %lld successes, %lld failures, %lld requested points.",
                      (long long)(pts.size() / 3), (long long)fail, (long long)num_points);
        *err = buf;
        return false;
    }
    return true;
}

}  // namespace c2b_host
