// Host side of noise.rs' index-corruption functions (src/noise.rs:179-378): add_incorrect_correspondences,
// drop_features, split_landmarks, join_landmarks.  They reshuffle the visibility graph's indices with sequential
// random draws -- no arithmetic worth a device -- so they run on the CPU over the flat CSR arrays.  The reference
// draws from thread_rng(); here one std::mt19937_64 per call, seeded by the caller.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <random>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace c2b_host {

inline double noise_uniform01(std::mt19937_64 &rng) { return std::uniform_real_distribution<double>(0.0, 1.0)(rng); }
inline uint64_t noise_below(std::mt19937_64 &rng, uint64_t n) { return std::uniform_int_distribution<uint64_t>(0, n - 1)(rng); }

// add_incorrect_correspondences, src/noise.rs:180-226.  Per camera with more than one observation, for each
// observation i in order: with probability mismatch_chance swap its point index with that of observation j drawn
// with weight w_k = (max_d - d_ik) for k != i -- and w_i = max_d, because the reference zeroes weights[i] BEFORE
// subtracting the minimum (:203-205), so the most likely partner is i itself (a no-op).  Image positions stay.
inline bool add_incorrect_correspondences(int64_t n_cam, const uint64_t *row_ptr, uint64_t *pt_idx, const double *uv,
                                          double mismatch_chance, uint64_t seed, std::string *err) {
    std::mt19937_64 rng(seed);
    std::vector<double> w;
    for (int64_t c = 0; c < n_cam; ++c) {
        const uint64_t b = row_ptr[c], len = row_ptr[c + 1] - b;
        if (len <= 1) continue;
        for (uint64_t i = 0; i < len; ++i) {
            if (!(noise_uniform01(rng) <= mismatch_chance)) continue;
            w.assign((size_t)len, 0.0);
            const double xi = uv[2 * (b + i)], yi = uv[2 * (b + i) + 1];
            for (uint64_t k = 0; k < len; ++k) {
                const double dx = xi - uv[2 * (b + k)], dy = yi - uv[2 * (b + k) + 1];
                w[(size_t)k] = -std::sqrt(dx * dx + dy * dy);
            }
            w[(size_t)i] = 0.0;
            double m = INFINITY;
            for (double x : w) m = std::min(m, x);
            double total = 0.0;
            for (double &x : w) { x -= m; total += x; }
            if (!(total > 0.0)) {                           // WeightedIndex::new(..).unwrap() panics: AllWeightsZero
                *err = "add_incorrect_correspondences: all swap weights are zero (camera " + std::to_string(c) + ")";
                return false;
            }
            // WeightedIndex::sample: first k with cumulative weight > U * total
            const double u = noise_uniform01(rng) * total;
            double cum = 0.0;
            uint64_t j = len - 1;
            for (uint64_t k = 0; k < len; ++k) {
                cum += w[(size_t)k];
                if (u < cum) { j = k; break; }
            }
            std::swap(pt_idx[b + i], pt_idx[b + j]);
        }
    }
    return true;
}

// drop_features, src/noise.rs:229-251: per camera shuffle the observations and keep the first
// floor(len * keep_fraction) (the argument is the fraction KEPT, :238-241).  Compacts the CSR arrays in place.
inline void drop_features(int64_t n_cam, uint64_t *row_ptr, uint64_t *pt_idx, double *uv, double keep_fraction, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<uint64_t> order;
    std::vector<uint64_t> pi;
    std::vector<double> xy;
    uint64_t w = 0, b = 0;
    for (int64_t c = 0; c < n_cam; ++c) {
        const uint64_t e = row_ptr[c + 1], len = e - b;
        double keep_f = (double)len * keep_fraction;
        uint64_t l = keep_f > 0.0 ? (keep_f >= (double)len ? len : (uint64_t)keep_f) : 0;    // `as usize` saturates
        order.resize((size_t)len);
        std::iota(order.begin(), order.end(), (uint64_t)0);
        for (uint64_t i = len; i > 1; --i) std::swap(order[(size_t)(i - 1)], order[(size_t)noise_below(rng, i)]);   // Fisher-Yates
        pi.resize((size_t)l); xy.resize((size_t)l * 2);
        for (uint64_t k = 0; k < l; ++k) {
            const uint64_t s = b + order[(size_t)k];
            pi[(size_t)k] = pt_idx[s]; xy[2 * (size_t)k] = uv[2 * s]; xy[2 * (size_t)k + 1] = uv[2 * s + 1];
        }
        for (uint64_t k = 0; k < l; ++k) {
            pt_idx[w + k] = pi[(size_t)k]; uv[2 * (w + k)] = xy[2 * (size_t)k]; uv[2 * (w + k) + 1] = xy[2 * (size_t)k + 1];
        }
        row_ptr[c] = w;      // row_ptr[c] (old start) is no longer needed: b carries the next old start
        w += l;
        b = e;
    }
    row_ptr[n_cam] = w;
    // row_ptr[c] above was overwritten with new starts while later iterations read row_ptr[c + 1] (old ends) only
}

// n distinct values out of [0, total), uniformly (IteratorRandom::choose_multiple): partial Fisher-Yates over a
// sparse permutation
inline std::vector<uint64_t> choose_multiple(uint64_t total, uint64_t n, std::mt19937_64 &rng) {
    std::vector<uint64_t> out;
    n = std::min(n, total);
    out.reserve((size_t)n);
    std::unordered_map<uint64_t, uint64_t> moved;
    auto at = [&](uint64_t i) { auto it = moved.find(i); return it == moved.end() ? i : it->second; };
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t j = k + noise_below(rng, total - k);
        const uint64_t vk = at(k), vj = at(j);
        out.push_back(vj);
        moved[j] = vk;
    }
    return out;
}

inline uint64_t fraction_of(double fraction, uint64_t count) {
    const double v = fraction * (double)count;
    return v > 0.0 ? (uint64_t)v : 0;
}

// split_landmarks, src/noise.rs:255-291: n = floor(split_fraction * n_pts) landmarks are duplicated (appended at
// n_pts .. n_pts + n in selection order); every observation of a selected landmark moves to its copy with
// probability 1/2.  pts3 must have room for n_pts + n rows.
inline int64_t split_landmarks(int64_t n_pts, double *pts3, int64_t n_obs, uint64_t *pt_idx, double split_fraction, uint64_t seed) {
    std::mt19937_64 rng(seed);
    const uint64_t n = std::min<uint64_t>(fraction_of(split_fraction, (uint64_t)n_pts), (uint64_t)n_pts);
    const std::vector<uint64_t> inds = choose_multiple((uint64_t)n_pts, n, rng);
    std::unordered_map<uint64_t, uint64_t> split;
    for (uint64_t k = 0; k < n; ++k) {
        for (int c = 0; c < 3; ++c) pts3[3 * ((uint64_t)n_pts + k) + c] = pts3[3 * inds[(size_t)k] + c];
        split[inds[(size_t)k]] = (uint64_t)n_pts + k;
    }
    for (int64_t o = 0; o < n_obs; ++o) {
        auto it = split.find(pt_idx[o]);
        if (it != split.end() && (rng() & 1u)) pt_idx[o] = it->second;
    }
    return n_pts + (int64_t)n;
}

// the 1 + k nearest points of pts[q] (itself included), ascending (distance, index): rstar's nearest_neighbor_iter
struct PointGrid {
    const double *pts;
    int64_t n;
    double lo[3], cell;
    int64_t dim[3];
    std::vector<int64_t> start, items;
    PointGrid(const double *p, int64_t n_pts) : pts(p), n(n_pts) {
        double hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int c = 0; c < 3; ++c) lo[c] = INFINITY;
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], p[3 * i + c]); hi[c] = std::max(hi[c], p[3 * i + c]); }
        double vol = 1.0, ext = 0.0;
        int live = 0;
        for (int c = 0; c < 3; ++c) { const double e = hi[c] - lo[c]; ext = std::max(ext, e); if (e > 0) { vol *= e; ++live; } }
        // about 4 points per cell over the live dimensions
        cell = live ? std::pow(vol * 4.0 / (double)std::max<int64_t>(n, 1), 1.0 / live) : 1.0;
        if (!(cell > 0.0) || !std::isfinite(cell)) cell = ext > 0 ? ext : 1.0;
        for (int c = 0; c < 3; ++c) {
            dim[c] = std::max<int64_t>(1, std::min<int64_t>(1024, (int64_t)std::floor((hi[c] - lo[c]) / cell) + 1));
        }
        start.assign((size_t)(dim[0] * dim[1] * dim[2] + 1), 0);
        for (int64_t i = 0; i < n; ++i) ++start[(size_t)key(p + 3 * i) + 1];
        for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
        items.resize((size_t)n);
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n; ++i) items[(size_t)fill[(size_t)key(p + 3 * i)]++] = i;
    }
    int64_t coord(const double *q, int c) const {
        return std::max<int64_t>(0, std::min(dim[c] - 1, (int64_t)std::floor((q[c] - lo[c]) / cell)));
    }
    int64_t key(const double *q) const { return (coord(q, 0) * dim[1] + coord(q, 1)) * dim[2] + coord(q, 2); }
    // ascending (d2, index) list of the k nearest points of q
    void nearest(const double *q, size_t k, std::vector<std::pair<double, int64_t>> &out) const {
        out.clear();
        const int64_t c0[3] = {coord(q, 0), coord(q, 1), coord(q, 2)};
        const int64_t rmax = std::max(dim[0], std::max(dim[1], dim[2]));
        for (int64_t r = 0; r <= rmax; ++r) {
            // shell r of cells around c0
            for (int64_t x = c0[0] - r; x <= c0[0] + r; ++x) {
                if (x < 0 || x >= dim[0]) continue;
                for (int64_t y = c0[1] - r; y <= c0[1] + r; ++y) {
                    if (y < 0 || y >= dim[1]) continue;
                    const bool face = std::llabs(x - c0[0]) == r || std::llabs(y - c0[1]) == r;
                    for (int64_t z = c0[2] - r; z <= c0[2] + r; z += (face ? 1 : std::max<int64_t>(1, 2 * r))) {
                        if (z < 0 || z >= dim[2]) continue;
                        const int64_t cellk = (x * dim[1] + y) * dim[2] + z;
                        for (int64_t s = start[(size_t)cellk]; s < start[(size_t)cellk + 1]; ++s) {
                            const int64_t i = items[(size_t)s];
                            const double d[3] = {pts[3 * i] - q[0], pts[3 * i + 1] - q[1], pts[3 * i + 2] - q[2]};
                            out.emplace_back((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], i);
                        }
                    }
                }
            }
            if (out.size() >= k) {
                std::sort(out.begin(), out.end());
                // everything not yet visited is farther than r * cell from q
                const double safe = (double)r * cell;
                if (out[k - 1].first <= safe * safe) break;
            }
        }
        std::sort(out.begin(), out.end());
        if (out.size() > k) out.resize(k);
    }
};

// join_landmarks, src/noise.rs:326-378: n = floor(join_fraction * n_pts) observations (distinct, uniform over all
// observations) get their landmark replaced by one of its 10 nearest other landmarks, chosen uniformly
// (nearest_neighbor_iter().skip(1).take(10).choose()).
inline bool join_landmarks(int64_t n_pts, const double *pts3, int64_t n_obs, uint64_t *pt_idx, double join_fraction,
                           uint64_t seed, std::string *err) {
    std::mt19937_64 rng(seed);
    const uint64_t n = fraction_of(join_fraction, (uint64_t)n_pts);
    const std::vector<uint64_t> inds = choose_multiple((uint64_t)n_obs, n, rng);
    if (inds.empty()) return true;
    if (n_pts < 2) { *err = "join_landmarks: No neighbors?!"; return false; }
    const PointGrid grid(pts3, n_pts);
    std::vector<std::pair<double, int64_t>> nn;
    for (uint64_t o : inds) {
        const uint64_t pi = pt_idx[o];
        grid.nearest(pts3 + 3 * pi, 11, nn);
        // skip(1): the nearest entry (the landmark itself unless an exact duplicate sorts before it)
        const size_t avail = nn.size() - 1;
        pt_idx[o] = (uint64_t)nn[1 + (size_t)noise_below(rng, avail)].second;
    }
    return true;
}

}  // namespace c2b_host
