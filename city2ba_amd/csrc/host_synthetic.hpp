// host_synthetic.hpp -- host-side (CPU, C++) pieces of the synthetic generators that surround the
// device predicate: layout of synthetic_grid / synthetic_line (src/synthetic.rs:178-258, 323-344),
// the candidate search that replaces rstar's locate_within_distance (:277-280, :362-365) and the
// 2-D building-occlusion test hits_building (:52-124).  None of this is the per-observation hot
// path ("next" row 3); it runs once per problem and feeds c2b_visibility_pairs.
//
// Third-party semantics restated (crates absent from the reference tree):
//   * rstar 0.7.1 locate_within_distance(p, r2): every point with squared distance <= r2, squared
//     distance accumulated in dimension order ((dx^2 + dy^2) + dz^2).  Its traversal order is not
//     reproducible without its source, so pairs are emitted in ascending point index per camera.
//   * line_intersection 0.4.0 LineInterval::relate (the "p + t r = q + u s" cross-product method):
//     parallel or collinear segments have no unique intersection; otherwise the intersection exists
//     iff 0 <= t <= 1 and 0 <= u <= 1 and is p + t r.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

namespace c2b_host {

// Basis3::from_angle_y(Deg(deg)), column-major; Deg -> Rad is deg * (PI / 180); libm sin/cos.
inline void basis_from_angle_y_deg(double deg, double *m) {
    const double th = deg * (3.14159265358979323846 / 180.0);
    const double s = std::sin(th), c = std::cos(th);
    const double t[9] = {c, 0.0, -s, 0.0, 1.0, 0.0, s, 0.0, c};
    std::copy(t, t + 9, m);
}

inline void grid_sizes(int64_t cpb, int64_t ppb, int64_t blocks, int64_t *n_cam, int64_t *n_pts) {
    *n_cam = 4 * cpb * blocks * (blocks + 1);
    *n_pts = 12 * ppb * blocks * (blocks + 1);
}

// src/synthetic.rs:178-258.  Writes camera positions [n][3], directions [n][9] (col-major) and
// points [n][3] in the reference's push order and arithmetic order.
inline void grid_layout(int64_t cpb, int64_t ppb, int64_t B, double L, double inset, double cam_h, double pt_h,
                        double *pos, double *dir, double *pts) {
    double d_m90[9], d_p90[9], d_180[9];
    const double d_one[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    basis_from_angle_y_deg(-90.0, d_m90);
    basis_from_angle_y_deg(90.0, d_p90);
    basis_from_angle_y_deg(180.0, d_180);
    auto push_cam = [&](double x, double y, double z, const double *d) {
        pos[0] = x; pos[1] = y; pos[2] = z; pos += 3;
        std::copy(d, d + 9, dir); dir += 9;
    };
    for (int64_t bx = 0; bx <= B; ++bx) {
        const double offset_x = L * (double)bx;
        for (int64_t by = 0; by <= B; ++by) {
            const double offset_z = L * (double)by;
            for (int64_t i = 0; i < cpb; ++i) {
                if (bx != B) {
                    const double x = offset_x + (double)i / (double)cpb * L;
                    push_cam(x, cam_h, offset_z, d_m90);
                    push_cam(x, cam_h, offset_z, d_p90);
                }
                if (by != B) {
                    const double z = offset_z + (double)i / (double)cpb * L;
                    push_cam(offset_x, cam_h, z, d_180);
                    push_cam(offset_x, cam_h, z, d_one);
                }
            }
        }
    }
    auto push_pt = [&](double x, double y, double z) { pts[0] = x; pts[1] = y; pts[2] = z; pts += 3; };
    for (int64_t bx = 0; bx <= B; ++bx) {
        const double offset_x = L * (double)bx;
        for (int64_t by = 0; by <= B; ++by) {
            const double offset_z = L * (double)by;
            for (int64_t i = 0; i < ppb; ++i) {
                const double step = (L - inset * 2.0) / (double)ppb;
                if (bx != B) {
                    const double loc_x = offset_x + inset + (double)i * step;
                    push_pt(loc_x, pt_h, offset_z - inset);
                    push_pt(loc_x, pt_h, offset_z + inset);
                    push_pt(loc_x + step / 2.0, 0.0, offset_z - inset);
                    push_pt(loc_x + step / 2.0, 0.0, offset_z + inset);
                    push_pt(loc_x + step / 2.0, 0.0, offset_z - inset / 2.0);
                    push_pt(loc_x + step / 2.0, 0.0, offset_z + inset / 2.0);
                }
                if (by != B) {
                    const double loc_z = offset_z + inset + (double)i * step;
                    push_pt(offset_x - inset, pt_h, loc_z);
                    push_pt(offset_x + inset, pt_h, loc_z);
                    push_pt(offset_x - inset, 0.0, loc_z + step / 2.0);
                    push_pt(offset_x + inset, 0.0, loc_z + step / 2.0);
                    push_pt(offset_x - inset / 2.0, 0.0, loc_z + step / 2.0);
                    push_pt(offset_x + inset / 2.0, 0.0, loc_z + step / 2.0);
                }
            }
        }
    }
}

// src/synthetic.rs:323-344
inline void line_layout(int64_t n_cam, int64_t n_pts, double length, double point_offset, double cam_h,
                        double pt_h, double *pos, double *dir, double *pts) {
    double d_180[9];
    basis_from_angle_y_deg(180.0, d_180);
    for (int64_t i = 0; i < n_cam; ++i) {
        pos[3 * i] = 0.0; pos[3 * i + 1] = cam_h;
        pos[3 * i + 2] = (double)i * length / (double)(n_cam - 1);
        std::copy(d_180, d_180 + 9, dir + 9 * i);
    }
    for (int64_t i = 0; i < n_pts; ++i) {
        const double z = (double)(i / 2) * length / (double)(n_pts / 2 - 1);
        pts[3 * i] = (i % 2 == 0) ? -point_offset : point_offset;
        pts[3 * i + 1] = pt_h;
        pts[3 * i + 2] = z;
    }
}

// ---- hits_building, src/synthetic.rs:52-124 ------------------------------------------------------
// unique intersection of segments (p0,p1) and (q0,q1); line_intersection 0.4.0 semantics
inline bool unique_intersection(double p0x, double p0y, double p1x, double p1y, double q0x, double q0y,
                                double q1x, double q1y, double *ix, double *iy) {
    const double rx = p1x - p0x, ry = p1y - p0y;
    const double sx = q1x - q0x, sy = q1y - q0y;
    const double r_cross_s = rx * sy - ry * sx;
    if (r_cross_s == 0.0) return false;                      // parallel or collinear
    const double qpx = q0x - p0x, qpy = q0y - p0y;
    const double t = qpx * (sy / r_cross_s) - qpy * (sx / r_cross_s);
    const double u = qpx * (ry / r_cross_s) - qpy * (rx / r_cross_s);
    if (!(0.0 <= t && t <= 1.0 && 0.0 <= u && u <= 1.0)) return false;
    *ix = p0x + t * rx;
    *iy = p0y + t * ry;
    return true;
}

inline bool hits_in_block(double sx, double sy, double ex, double ey, int64_t bix, int64_t biy, double L,
                          double inset) {
    const double block_end = L - inset;
    const double ox = (double)bix * L, oy = (double)biy * L;
    const double sides[4][4] = {
        {ox + inset, oy + inset, ox + inset, oy + block_end},
        {ox + inset, oy + inset, ox + block_end, oy + inset},
        {ox + block_end, oy + inset, ox + block_end, oy + block_end},
        {ox + inset, oy + block_end, ox + block_end, oy + block_end},
    };
    for (int k = 0; k < 4; ++k) {
        double px, py;
        // view_segment.relate(&side_segment)
        if (unique_intersection(sx, sy, ex, ey, sides[k][0], sides[k][1], sides[k][2], sides[k][3], &px, &py)) {
            // reference quirk kept (src/synthetic.rs:93): the y term is NOT squared, so the radicand can
            // be negative -> sqrt = NaN -> comparison false -> "no hit"
            const double dx = ex - px;
            if (std::sqrt(dx * dx + (ey - py)) > 1e-8) return true;
        }
    }
    return false;
}

inline bool hits_building(const double c[3], const double p[3], double L, double inset) {
    const double sx = c[0], sy = c[2], ex = p[0], ey = p[2];
    const int64_t cbx = (int64_t)std::trunc(sx / L), cby = (int64_t)std::trunc(sy / L);
    const int64_t pbx = (int64_t)std::trunc(ex / L), pby = (int64_t)std::trunc(ey / L);
    const int64_t x0 = std::min(cbx, pbx), x1 = std::max(cbx, pbx);
    const int64_t y0 = std::min(cby, pby), y1 = std::max(cby, pby);
    for (int64_t bx = x0; bx <= x1; ++bx)
        for (int64_t by = y0; by <= y1; ++by)
            if (hits_in_block(sx, sy, ex, ey, bx, by, L, inset)) return true;
    return false;
}

// ---- candidate search ----------------------------------------------------------------------------
struct Pairs {
    std::vector<uint32_t> cam, pt;
};

// All (camera, point) pairs, cameras [cam_lo, cam_hi), with squared distance <= max_dist^2, optionally
// dropping pairs whose sight line hits a building.  Camera-major, ascending point index per camera.
inline void candidate_pairs(const double *centers, const double *pts, int64_t n_pts, double max_dist,
                            int64_t cam_lo, int64_t cam_hi, bool occlusion, double L, double inset,
                            int n_threads, Pairs *out) {
    out->cam.clear();
    out->pt.clear();
    if (cam_hi <= cam_lo || n_pts == 0) return;
    const double cs = max_dist > 0 ? max_dist : 1.0;
    const double r2 = max_dist * max_dist;
    double x0 = pts[0], z0 = pts[2], x1 = pts[0], z1 = pts[2];
    for (int64_t j = 0; j < n_pts; ++j) {
        x0 = std::min(x0, pts[3 * j]); x1 = std::max(x1, pts[3 * j]);
        z0 = std::min(z0, pts[3 * j + 2]); z1 = std::max(z1, pts[3 * j + 2]);
    }
    const int64_t ncx = (int64_t)std::floor((x1 - x0) / cs) + 1, ncz = (int64_t)std::floor((z1 - z0) / cs) + 1;
    auto cell_of = [&](double x, double z, int64_t *cx, int64_t *cz) {
        *cx = (int64_t)std::floor((x - x0) / cs);
        *cz = (int64_t)std::floor((z - z0) / cs);
    };
    // counting sort of points by cell (stable: ascending index inside a cell)
    std::vector<int64_t> start((size_t)(ncx * ncz + 1), 0);
    std::vector<uint32_t> pcell((size_t)n_pts);
    for (int64_t j = 0; j < n_pts; ++j) {
        int64_t cx, cz;
        cell_of(pts[3 * j], pts[3 * j + 2], &cx, &cz);
        pcell[(size_t)j] = (uint32_t)(cx * ncz + cz);
        ++start[(size_t)pcell[(size_t)j] + 1];
    }
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    std::vector<uint32_t> sorted((size_t)n_pts);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t j = 0; j < n_pts; ++j) sorted[(size_t)fill[pcell[(size_t)j]]++] = (uint32_t)j;
    }
    if (n_threads < 1) n_threads = 1;
    const int64_t n = cam_hi - cam_lo;
    if ((int64_t)n_threads > n) n_threads = (int)n;
    std::vector<Pairs> part((size_t)n_threads);
    auto work = [&](int t) {
        const int64_t lo = cam_lo + n * t / n_threads, hi = cam_lo + n * (t + 1) / n_threads;
        Pairs &P = part[(size_t)t];
        std::vector<uint32_t> cand;
        for (int64_t c = lo; c < hi; ++c) {
            const double *ctr = centers + 3 * c;
            int64_t ccx, ccz;
            cell_of(ctr[0], ctr[2], &ccx, &ccz);
            cand.clear();
            for (int64_t cx = ccx - 1; cx <= ccx + 1; ++cx) {
                if (cx < 0 || cx >= ncx) continue;
                for (int64_t cz = ccz - 1; cz <= ccz + 1; ++cz) {
                    if (cz < 0 || cz >= ncz) continue;
                    const int64_t id = cx * ncz + cz;
                    for (int64_t k = start[(size_t)id]; k < start[(size_t)id + 1]; ++k) {
                        const uint32_t j = sorted[(size_t)k];
                        const double dx = ctr[0] - pts[3 * (int64_t)j], dy = ctr[1] - pts[3 * (int64_t)j + 1],
                                     dz = ctr[2] - pts[3 * (int64_t)j + 2];
                        if ((dx * dx + dy * dy) + dz * dz <= r2) cand.push_back(j);
                    }
                }
            }
            std::sort(cand.begin(), cand.end());
            for (uint32_t j : cand) {
                if (occlusion && hits_building(ctr, pts + 3 * (int64_t)j, L, inset)) continue;
                P.cam.push_back((uint32_t)c);
                P.pt.push_back(j);
            }
        }
    };
    run_threads(n_threads, work);
    size_t total = 0;
    for (auto &P : part) total += P.cam.size();
    out->cam.reserve(total);
    out->pt.reserve(total);
    for (auto &P : part) {
        out->cam.insert(out->cam.end(), P.cam.begin(), P.cam.end());
        out->pt.insert(out->pt.end(), P.pt.begin(), P.pt.end());
    }
}

}  // namespace c2b_host
