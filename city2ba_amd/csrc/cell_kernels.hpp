// cell_kernels.hpp -- the generators' visibility loop on the device, candidates included (included by capi.hip only).
//
// synthetic_grid / synthetic_line (src/synthetic.rs:268-297, :353-378) run, per camera (rayon par_iter over cameras):
//     for p in rtree.locate_within_distance(camera.center(), max_dist * max_dist)       -- rstar: squared distance <= r2
//         if !hits_building(camera.center(), p)                                           -- grid only (:52-124)
//             if (camera.center() - p).magnitude() < max_dist && project_world(p).z <= 0 && -1 <= u, v <= 1  -> push
// Rounds 1-3 did the R-tree query and hits_building on the host (290 ms of a 1.2 s `synthetic --blocks 128`) and shipped
// 47.5 M candidate pairs to the device for the predicate.  Here the whole loop runs on the device:
//   * the R-tree is replaced by a uniform cell list over the (x, z) plane with cells at least max_dist wide: every point
//     within max_dist of a camera lies in the 3 x 3 cells around the camera's cell, and the three cells of one column
//     are one contiguous range of the cell-sorted point list;
//   * one wave per camera walks those three ranges 64 points at a time: squared distance, projection + predicate, then
//     (grid) the sight line against the buildings -- the cheap tests first, the result is their conjunction;
//   * two passes (count, scan, fill) like the dense sweep; the fill writes a camera's survivors in the order the wave met
//     them and k_rows_rank_sort puts every row into ASCENDING POINT INDEX, the canonical in-camera order of this build
//     (rstar's traversal order is not reproducible, host_synthetic.hpp) -- so the result does not depend on the order the
//     atomics of the cell fill happened to produce.
// Everything the reference computes in f64 (distances, the intersection arithmetic of hits_building, the projection) is
// evaluated in the same operation order as csrc/host_synthetic.hpp and camera_math.hpp: kept indices are bit-equal to the
// host path's.
#pragma once
#include "camera_math.hpp"

namespace c2b {

struct CellGrid {
    double x0, z0, inv_cs;       // cell (i, k) covers x0 + [i, i + 1) cs, z0 + [k, k + 1) cs
    int ncx, ncz;
};

C2B_DEV int cell_coord(double v, double v0, double inv_cs, int n) {
    const double f = floor((v - v0) * inv_cs);
    return f < 0.0 ? 0 : (f >= (double)n ? n - 1 : (int)f);       // clamped (NaN converts to cell 0): conservative, the distance test decides
}

// cell of every point + points per cell (integer atomics: the counts do not depend on their order)
__global__ __launch_bounds__(256) void k_cells_assign(const double4 *__restrict__ pts4, int64_t n, CellGrid g,
                                                     uint32_t *__restrict__ cell_of, uint32_t *__restrict__ counts) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double4 p = pts4[j];
    const uint32_t c = (uint32_t)(cell_coord(p.x, g.x0, g.inv_cs, g.ncx) * g.ncz + cell_coord(p.z, g.z0, g.inv_cs, g.ncz));
    cell_of[j] = c;
    atomicAdd(counts + c, 1u);
}

// the cell-sorted point list (order inside a cell = arrival order of the atomics; k_rows_rank_sort makes the result
// independent of it)
__global__ __launch_bounds__(256) void k_cells_fill(const uint32_t *__restrict__ cell_of, int64_t n, const uint32_t *__restrict__ start,
                                                   uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t c = cell_of[j];
    sorted[start[c] + atomicAdd(cursor + c, 1u)] = (uint32_t)j;
}

// ---- hits_building, src/synthetic.rs:52-124, as csrc/host_synthetic.hpp restates it (same operations, same order) ----
// unique intersection of segments (p0, p1) and (q0, q1): line_intersection 0.4.0's relate().unique_intersection()
C2B_DEV bool seg_unique_intersection(double p0x, double p0y, double p1x, double p1y, double q0x, double q0y, double q1x,
                                     double q1y, double &ix, double &iy) {
    const double rx = p1x - p0x, ry = p1y - p0y;
    const double sx = q1x - q0x, sy = q1y - q0y;
    const double r_cross_s = rx * sy - ry * sx;
    if (r_cross_s == 0.0) return false;                      // parallel or collinear
    const double qpx = q0x - p0x, qpy = q0y - p0y;
    const double t = qpx * (sy / r_cross_s) - qpy * (sx / r_cross_s);
    const double u = qpx * (ry / r_cross_s) - qpy * (rx / r_cross_s);
    if (!(0.0 <= t && t <= 1.0 && 0.0 <= u && u <= 1.0)) return false;
    ix = p0x + t * rx;
    iy = p0y + t * ry;
    return true;
}

C2B_DEV bool hits_in_block_dev(double sx, double sy, double ex, double ey, int64_t bix, int64_t biy, double L, double inset) {
    const double block_end = L - inset;
    const double ox = (double)bix * L, oy = (double)biy * L;
    const double lo_x = ox + inset, lo_y = oy + inset, hi_x = ox + block_end, hi_y = oy + block_end;
    // the four sides in the reference's order (:72-89); any hit ends the test, so the order only matters for speed
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double ax = k == 2 ? hi_x : lo_x, ay = k == 3 ? hi_y : lo_y;
        const double bx = k == 0 ? lo_x : hi_x, by = k == 1 ? lo_y : hi_y;
        double px, py;
        if (seg_unique_intersection(sx, sy, ex, ey, ax, ay, bx, by, px, py)) {
            // reference quirk kept (src/synthetic.rs:93): the y term is NOT squared, so the radicand can be negative
            // -> sqrt = NaN -> comparison false -> "no hit"
            const double dx = ex - px;
            if (sqrt(dx * dx + (ey - py)) > 1e-8) return true;
        }
    }
    return false;
}

C2B_DEV bool hits_building_dev(double cx, double cz, double px, double pz, double L, double inset) {
    // block index by truncation (:104-105), wrong for negative coordinates like the reference
    const int64_t cbx = (int64_t)trunc(cx / L), cby = (int64_t)trunc(cz / L);
    const int64_t pbx = (int64_t)trunc(px / L), pby = (int64_t)trunc(pz / L);
    const int64_t x0 = cbx < pbx ? cbx : pbx, x1 = cbx < pbx ? pbx : cbx;
    const int64_t y0 = cby < pby ? cby : pby, y1 = cby < pby ? pby : cby;
    for (int64_t bx = x0; bx <= x1; ++bx)
        for (int64_t by = y0; by <= y1; ++by)
            if (hits_in_block_dev(cx, cz, px, pz, bx, by, L, inset)) return true;
    return false;
}

// One wave per camera.  FILL = false: cam_count[c] = survivors of camera c.  FILL = true: survivors (point index, uv)
// at row_ptr[c] ..., in the order the wave meets them.
constexpr int kCellWPB = 4;
template <bool FILL>
__global__ __launch_bounds__(kCellWPB * 64) void k_cells_visibility(
    const double *__restrict__ camblk, int64_t n_cam, const double4 *__restrict__ pts4, CellGrid g,
    const uint32_t *__restrict__ start, const uint32_t *__restrict__ sorted, double max_dist, int occlusion, double L,
    double inset, uint32_t *__restrict__ cam_count, const uint64_t *__restrict__ row_ptr, uint32_t *__restrict__ pt_out,
    double2 *__restrict__ uv_out) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * kCellWPB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (c >= n_cam) return;                                              // wave-uniform; no workgroup barrier below
    const CamRec cam(camblk, c);                                         // wave-uniform addresses: scalar loads
    const double gx = cam[kCenter], gy = cam[kCenter + 1], gz = cam[kCenter + 2];
    const int ccx = cell_coord(gx, g.x0, g.inv_cs, g.ncx), ccz = cell_coord(gz, g.z0, g.inv_cs, g.ncz);
    const double r2 = max_dist * max_dist;
    uint64_t dst = FILL ? row_ptr[c] : 0;
    uint32_t count = 0;
    const int ix0 = ccx > 0 ? ccx - 1 : 0, ix1 = ccx + 1 < g.ncx ? ccx + 1 : g.ncx - 1;
    const int iz0 = ccz > 0 ? ccz - 1 : 0, iz1 = ccz + 1 < g.ncz ? ccz + 1 : g.ncz - 1;
    for (int ix = ix0; ix <= ix1; ++ix) {
        const uint32_t b = start[ix * g.ncz + iz0], e = start[ix * g.ncz + iz1 + 1];      // three cells, one range
        for (uint32_t k0 = b; k0 < e; k0 += 64) {                                          // wave-uniform trip count
            const uint32_t k = k0 + lane;
            const bool valid = k < e;
            const uint32_t j = sorted[valid ? k : e - 1];
            const double4 X = pts4[j];
            // rstar's locate_within_distance: squared distance <= max_dist^2, accumulated in dimension order
            const double dx = gx - X.x, dy = gy - X.y, dz = gz - X.z;
            const double d2 = dot3(dx, dy, dz, dx, dy, dz);
            bool keep = valid && d2 <= r2;
            double u = 0.0, v = 0.0;
            if (keep) {
                // (center - p).magnitude() < max_dist && q.z <= 0 && -1 <= u, v <= 1   (src/synthetic.rs:285-291, :368-375)
                const Proj p = project_obs(cam, X.x, X.y, X.z);
                u = p.u; v = p.v;
                keep = sqrt(d2) < max_dist && p.qz <= 0.0 && u >= -1.0 && u <= 1.0 && v >= -1.0 && v <= 1.0;
            }
            if (keep && occlusion) keep = !hits_building_dev(gx, gz, X.x, X.z, L, inset);
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            if (FILL && keep) {
                const uint64_t at = dst + (uint64_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                pt_out[at] = j;
                uv_out[at] = make_double2(u, v);
            }
            const uint32_t got = (uint32_t)__builtin_popcountll(m);
            count += got;
            dst += got;
        }
    }
    if (!FILL && lane == 0) cam_count[c] = count;
}

// Every row into ascending point index: entry i goes to position #{entries of its row with a smaller index} (indices are
// distinct inside a row).  One wave per camera; rows are a few dozen entries (the grid: 29 on average), all in L1.
__global__ __launch_bounds__(kCellWPB * 64) void k_rows_rank_sort(const uint64_t *__restrict__ row_ptr, int64_t n_cam,
                                                                const uint32_t *__restrict__ pt_in, const double2 *__restrict__ uv_in,
                                                                uint32_t *__restrict__ pt_out, double2 *__restrict__ uv_out) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * kCellWPB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (c >= n_cam) return;
    const uint64_t b = row_ptr[c], e = row_ptr[c + 1];
    for (uint64_t i0 = b; i0 < e; i0 += 64) {                            // wave-uniform trip count
        const uint64_t i = i0 + lane;
        const bool valid = i < e;
        const uint32_t mine = pt_in[valid ? i : e - 1];
        uint32_t rank = 0;
        for (uint64_t j = b; j < e; ++j) rank += pt_in[j] < mine ? 1u : 0u;
        if (valid) {
            pt_out[b + rank] = mine;
            uv_out[b + rank] = uv_in[i];
        }
    }
}

// largest of n u32 counts (the longest row: k_rows_rank_sort is quadratic in it)
__global__ __launch_bounds__(256) void k_max_u32(const uint32_t *__restrict__ v, int64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = v[i] > m ? v[i] : m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// sum of n u32 counts in 64 bits (the scans run in 32 bits: this catches a total that does not fit them)
__global__ __launch_bounds__(256) void k_sum_u32_u64(const uint32_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// ---- the layout loops of synthetic_grid / synthetic_line (src/synthetic.rs:178-258, :323-344) on the device ---------
// One thread per camera / point: the loop nest's push order is inverted in closed form (block column bx, block row by,
// slot i, which of the block's entities) and every coordinate is evaluated with the reference's operations in its order
// (`offset + i as f64 / n as f64 * L`, `(L - inset * 2.) / n`, ...), so positions are bit-equal to the host loops of
// csrc/host_synthetic.hpp.  The four camera orientations (Basis3::from_angle_y of -90, 90, 180 degrees and the identity,
// :191-205) are evaluated by the host's libm once and passed in; cameras are finished here like
// Camera::from_position_direction (src/baproblem.rs:153-159).
struct GridDirs { double m[4][9]; };          // column-major: from_angle_y(-90), (90), (180), one

// entity n of a loop nest that pushes `per` entities per slot for each of the two street directions:
// for bx in 0..=B { for by in 0..=B { for i in 0..slots { if bx != B { per x } if by != B { per x } } } }
C2B_DEV void grid_slot(int64_t n, int64_t B, int64_t slots, int per, int64_t &bx, int64_t &by, int64_t &i, int &k, bool &along_x) {
    const int64_t row_full = slots * per * (2 * B + 1);                  // a column bx < B: B blocks of 2 per, one of per
    if (n < B * row_full) {
        bx = n / row_full;
        const int64_t r = n % row_full;
        if (r < 2 * per * slots * B) {
            by = r / (2 * per * slots);
            const int64_t r2 = r % (2 * per * slots);
            i = r2 / (2 * per);
            const int kk = (int)(r2 % (2 * per));
            along_x = kk < per; k = kk % per;
        } else {
            by = B;
            const int64_t r2 = r - 2 * per * slots * B;
            i = r2 / per; k = (int)(r2 % per); along_x = true;
        }
    } else {                                                              // the last column: only the streets along z
        bx = B;
        const int64_t r = n - B * row_full;
        by = r / (per * slots);
        const int64_t r2 = r % (per * slots);
        i = r2 / per; k = (int)(r2 % per); along_x = false;
    }
}

__global__ __launch_bounds__(256) void k_grid_cameras(int64_t n_cam, int64_t cpb, int64_t B, double L, double cam_h, GridDirs d,
                                                     double *__restrict__ cam15) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_cam) return;
    int64_t bx, by, i;
    int k;
    bool along_x;
    grid_slot(n, B, cpb, 2, bx, by, i, k, along_x);
    const double offset_x = L * (double)bx, offset_z = L * (double)by;
    const double t = (double)i / (double)cpb * L;
    const double px = along_x ? offset_x + t : offset_x, pz = along_x ? offset_z : offset_z + t;
    const double *R = d.m[(along_x ? 0 : 2) + k];
    double v[3];
    cm_mat_vec(R, px, cam_h, pz, v);
    double *o = cam15 + 15 * n;
#pragma unroll
    for (int q = 0; q < 9; ++q) o[q] = R[q];
    o[9] = -1.0 * v[0]; o[10] = -1.0 * v[1]; o[11] = -1.0 * v[2];
    o[12] = 1.0; o[13] = 0.0; o[14] = 0.0;
}

__global__ __launch_bounds__(256) void k_grid_points(int64_t n_pts, int64_t ppb, int64_t B, double L, double inset, double pt_h,
                                                    double4 *__restrict__ pts4) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_pts) return;
    int64_t bx, by, i;
    int k;
    bool along_x;
    grid_slot(n, B, ppb, 6, bx, by, i, k, along_x);
    const double offset_x = L * (double)bx, offset_z = L * (double)by;
    const double step = (L - inset * 2.0) / (double)ppb;
    // the six points of a slot (:219-253): two at point_height on either side of the street, four on the ground
    const double loc = (along_x ? offset_x : offset_z) + inset + (double)i * step;
    const double a = k < 2 ? loc : loc + step / 2.0;                      // along the street
    const double o = along_x ? offset_z : offset_x;                       // across it
    const double half = inset / 2.0;
    const double c = k == 0 || k == 2 ? o - inset : (k == 1 || k == 3 ? o + inset : (k == 4 ? o - half : o + half));
    const double y = k < 2 ? pt_h : 0.0;
    pts4[n] = along_x ? make_double4(a, y, c, 0.0) : make_double4(c, y, a, 0.0);
}

__global__ __launch_bounds__(256) void k_line_layout(int64_t n_cam, int64_t n_pts, double length, double point_offset, double cam_h,
                                                    double pt_h, GridDirs d, double *__restrict__ cam15, double4 *__restrict__ pts4) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < n_cam) {
        const double *R = d.m[2];                                        // from_angle_y(Deg(180)), :330
        double v[3];
        cm_mat_vec(R, 0.0, cam_h, (double)n * length / (double)(n_cam - 1), v);
        double *o = cam15 + 15 * n;
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q] = R[q];
        o[9] = -1.0 * v[0]; o[10] = -1.0 * v[1]; o[11] = -1.0 * v[2];
        o[12] = 1.0; o[13] = 0.0; o[14] = 0.0;
    }
    if (n < n_pts) {
        const double z = (double)(n / 2) * length / (double)(n_pts / 2 - 1);
        pts4[n] = make_double4((n % 2 == 0) ? -point_offset : point_offset, pt_h, z, 0.0);
    }
}

// ---- BAProblem::write_binary (src/baproblem.rs:736-764) assembled on the device -------------------------------------
// The .bbal image: header (3 big-endian u64: cameras, points, observations) | per camera: BE u64 count, then count x
// (BE u64 point, BE f64 u, BE f64 v) | 9 BE f64 per camera (to_vec order) | 3 BE f64 per point.  Every word is 8-byte
// aligned in the file, so the image is an array of u64 and each kernel below writes byte-swapped words into it; the host
// only moves bytes (pinned chunks -> pwrite).  463 MB of a 564 MB file at --blocks 128 are observation records.
C2B_DEV uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }

__global__ __launch_bounds__(256) void k_bbal_rows(const uint64_t *__restrict__ row_ptr, int64_t n_cam, int64_t n_pts, int64_t n_obs,
                                                  uint64_t *__restrict__ img) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c == 0) { img[0] = bswap64((uint64_t)n_cam); img[1] = bswap64((uint64_t)n_pts); img[2] = bswap64((uint64_t)n_obs); }
    if (c >= n_cam) return;
    const uint64_t b = row_ptr ? row_ptr[c] : 0, e = row_ptr ? row_ptr[c + 1] : 0;     // no row structure: no observations
    img[3 + c + 3 * b] = bswap64(e - b);                                 // its count sits in front of its records
}
__global__ __launch_bounds__(256) void k_bbal_observations(const uint32_t *__restrict__ cam_idx, const uint32_t *__restrict__ pt_idx,
                                                          const double2 *__restrict__ uv, int64_t n_obs, uint64_t *__restrict__ img) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obs) return;
    const double2 q = uv[o];
    uint64_t *w = img + 3 + ((uint64_t)cam_idx[o] + 1) + 3 * (uint64_t)o;   // header | one count per camera up to and including its own
    w[0] = bswap64((uint64_t)pt_idx[o]);
    w[1] = bswap64((uint64_t)__double_as_longlong(q.x));
    w[2] = bswap64((uint64_t)__double_as_longlong(q.y));
}
// n rows of `width` doubles out of rows `stride` doubles apart (cameras: 9 of 9; points: 3 of the padded 4)
__global__ __launch_bounds__(256) void k_bbal_rows_f64(const double *__restrict__ in, int64_t n, int width, int stride,
                                                      uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * width) return;
    const int64_t r = i / width, k = i % width;
    out[i] = bswap64((uint64_t)__double_as_longlong(in[r * stride + k]));
}

// ---- BAProblem::from_file_binary (src/baproblem.rs:632-695) decoded on the device ------------------------------------
// `raw` = the file's bytes as u64 words; the host has walked the per-camera counts (each sits in front of its records, so
// finding them is a pointer chase through the file) and uploaded row_ptr; everything per observation happens here.
__global__ __launch_bounds__(256) void k_bbal_read_observations(const uint64_t *__restrict__ raw, const uint32_t *__restrict__ cam_idx,
                                                               int64_t n_obs, uint64_t n_pts, uint32_t *__restrict__ pt_idx,
                                                               double2 *__restrict__ uv, uint32_t *__restrict__ bad) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obs) return;
    const uint64_t *w = raw + 3 + ((uint64_t)cam_idx[o] + 1) + 3 * (uint64_t)o;
    const uint64_t pt = bswap64(w[0]);
    if (pt >= n_pts) atomicOr(bad, 1u);                                  // assert!(ci < &points.len()), src/baproblem.rs:368
    pt_idx[o] = (uint32_t)pt;
    uv[o] = make_double2(__longlong_as_double((long long)bswap64(w[1])), __longlong_as_double((long long)bswap64(w[2])));
}
// n rows of `width` big-endian doubles -> rows `stride` doubles apart (cameras 9 -> 9, points 3 -> the padded 4)
__global__ __launch_bounds__(256) void k_bbal_read_rows_f64(const uint64_t *__restrict__ in, int64_t n, int width, int stride,
                                                           double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * stride) return;
    const int64_t r = i / stride, k = i % stride;
    out[i] = k < width ? __longlong_as_double((long long)bswap64(in[r * width + k])) : 0.0;
}

// ---- generate_world_points_uniform (src/generate.rs:356-420) on the device ---------------------------------------------
// The reference's loop: draw a triangle by area, a point in it, keep it if some camera lies within max_dist (an rstar
// query), stop at num_points successes or 10 * num_points failures.  csrc/host_generate.hpp gives candidate k its own
// splitmix64 stream and accepts candidates in order; these kernels evaluate the same candidates -- the same stream, the
// same operations in the same order, the same `<=` on the squared distance -- so the accepted points are the host's,
// bit for bit.  "Some camera within max_dist" is decided through the cell list over the camera centres (cells at least
// max_dist wide: such a camera sits in the 3 x 3 cells around the point's, and the clamp of cell_coord never hides one).
struct SplitMix {                                            // host_generate.hpp: CounterRng
    uint64_t s;
    C2B_DEV SplitMix(uint64_t seed, uint64_t k) : s(seed * 0xD6E8FEB86659FD93ull + k * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull) {}
    C2B_DEV uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    C2B_DEV double uniform01() { return (double)(next() >> 11) * 0x1.0p-53; }
};

// candidates k0 .. k0 + chunk: cand[i] = the point, ok[i] = 1 if a camera is near
__global__ __launch_bounds__(256) void k_world_candidates(const float *__restrict__ tri9, int64_t n_tri, const double *__restrict__ cum,
                                                         uint64_t seed, int64_t k0, int64_t chunk, const double4 *__restrict__ centres,
                                                         CellGrid g, const uint32_t *__restrict__ start, const uint32_t *__restrict__ sorted,
                                                         double max_dist, double4 *__restrict__ cand, uint32_t *__restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunk) return;
    SplitMix rng(seed, (uint64_t)(k0 + i));
    // WeightedIndex::at: the first cumulative area above u (std::upper_bound), clamped to the last triangle
    const double u = rng.uniform01() * cum[n_tri - 1];
    int64_t lo = 0, hi = n_tri;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (!(u < cum[mid])) lo = mid + 1; else hi = mid;
    }
    const int64_t t = lo < n_tri - 1 ? lo : n_tri - 1;
    const float *q = tri9 + 9 * t;
    double rx = rng.uniform01(), ry = rng.uniform01();
    if (rx + ry > 1.0) { rx = 1.0 - rx; ry = 1.0 - ry; }
    double p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = (double)q[c] + rx * ((double)q[3 + c] - (double)q[c]) + ry * ((double)q[6 + c] - (double)q[c]);
    cand[i] = make_double4(p[0], p[1], p[2], 0.0);
    const int cx = cell_coord(p[0], g.x0, g.inv_cs, g.ncx), cz = cell_coord(p[2], g.z0, g.inv_cs, g.ncz);
    const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz + 1 < g.ncz ? cz + 1 : g.ncz - 1;
    const double r2 = max_dist * max_dist;
    bool near = false;
    for (int x = (cx > 0 ? cx - 1 : 0); x <= (cx + 1 < g.ncx ? cx + 1 : g.ncx - 1) && !near; ++x) {
        const uint32_t a = start[x * g.ncz + z0], b = start[x * g.ncz + z1 + 1];      // three cells of a column: one range
        for (uint32_t j = a; j < b; ++j) {
            const double4 c = centres[sorted[j]];
            const double e0 = c.x - p[0], e1 = c.y - p[1], e2 = c.z - p[2];
            if ((e0 * e0 + e1 * e1) + e2 * e2 <= r2) { near = true; break; }
        }
    }
    ok[i] = near ? 1u : 0u;
}

// The sequential loop takes candidate i while fewer than `need` were accepted and fewer than `fails_left` failed before
// it: cutoff = the first candidate it does not take (chunk if it takes them all).  pos = exclusive scan of ok.
__global__ __launch_bounds__(256) void k_world_cutoff(const uint32_t *__restrict__ pos, int64_t chunk, uint64_t need, uint64_t fails_left,
                                                     unsigned long long *__restrict__ cutoff) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunk) return;
    const uint64_t acc = pos[i], failed = (uint64_t)i - acc;
    if (acc >= need || failed >= fails_left) atomicMin(cutoff, (unsigned long long)i);
}
__global__ __launch_bounds__(256) void k_world_accept(const double4 *__restrict__ cand, const uint32_t *__restrict__ ok,
                                                     const uint32_t *__restrict__ pos, int64_t chunk,
                                                     const unsigned long long *__restrict__ cutoff, double4 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunk || (unsigned long long)i >= *cutoff || !ok[i]) return;
    out[pos[i]] = cand[i];
}

}  // namespace c2b
