/*
 * city2ba_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's CPU algorithm for the hot path
 * (tkonolige/city2ba v1.1.1, Rust).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared object; the product
 * (city2ba_amd/, include/) never links, imports or calls it.
 *
 * Parity status ("how pinned is this oracle?")
 *   - The reference is Rust and cannot be built here (no rustc/cargo, no
 *     libembree3), so no reference-run fixtures exist.  The arithmetic lives
 *     partly in un-vendored crates: cgmath 0.17.0 (Cargo.lock), rand 0.6.5.
 *     cgmath's published algorithms (column-major Matrix3, axis-angle matrix,
 *     trace-method quaternion, cofactor inverse, v * (1/|v|) normalisation)
 *     are restated below op-for-op from knowledge of that release.
 *   - PINNED by the reference's own known-answer tests for this path
 *     (src/baproblem.rs:64-75, 227-249: rodrigues_idempotent,
 *     test_project_world, test_project, test_project_isomorphic), by the
 *     zero-error-by-construction invariant of the generators, by the eight
 *     inequalities of tests/main.rs:134-195, and by mpmath 50-digit golden
 *     vectors of the BAL/Snavely model (tests/golden/).
 *   - PARITY UNPINNED (no reference implementation or vector exists):
 *       * the 2x(9+3) Jacobian (absent from the reference; derived here and
 *         checked against mpmath analytic + central differences);
 *       * every noise draw (the reference uses unseeded rand::thread_rng();
 *         this oracle defines a Philox4x32-10 counter scheme of its own and
 *         restates only the deterministic algebra around the draws).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; Rust never
 * contracts a*b+c into an FMA, so neither may this file).
 *
 * Conventions
 *   mat3 is COLUMN-MAJOR, m[3*c + r], exactly like cgmath::Matrix3 whose
 *   Index<usize> yields columns (mat[c][r]).
 *   cam15 = { R col-major [9], t (loc) [3], f, k1, k2 } : the in-memory
 *   SnavelyCamera {dir, loc, intrin} of src/baproblem.rs:130-138.
 *   bal9  = { w0,w1,w2, t0,t1,t2, f,k1,k2 } : SnavelyCamera::to_vec order,
 *   src/baproblem.rs:189-202.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define M(m, c, r) ((m)[3 * (c) + (r)])

/* ------------------------------------------------------------------ */
/* cgmath 0.17 primitives (un-vendored; restated from the published    */
/* source of that release).                                            */
/* ------------------------------------------------------------------ */

/* Vector3::dot = mul_element_wise().sum() = (x*x' + y*y') + z*z' */
static double dot3(const double a[3], const double b[3]) {
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* InnerSpace::magnitude = sqrt(dot(self,self)) */
static double mag3(const double a[3]) { return sqrt(dot3(a, a)); }
/* InnerSpace::normalize = self * (1 / magnitude) */
static void normalize3(const double a[3], double out[3]) {
    double s = 1.0 / mag3(a);
    out[0] = a[0] * s; out[1] = a[1] * s; out[2] = a[2] * s;
}
/* Vector3::cross */
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* Matrix3 * Vector3 = (row0.dot(v), row1.dot(v), row2.dot(v)) */
static void mat_vec(const double m[9], const double v[3], double o[3]) {
    for (int r = 0; r < 3; ++r) {
        double row[3] = { M(m, 0, r), M(m, 1, r), M(m, 2, r) };
        o[r] = dot3(row, v);
    }
}
/* Matrix3 * Matrix3: out[c][r] = lhs.row(r).dot(rhs[c]) */
static void mat_mul(const double a[9], const double b[9], double o[9]) {
    double t[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double row[3] = { M(a, 0, r), M(a, 1, r), M(a, 2, r) };
            t[3 * c + r] = dot3(row, &b[3 * c]);
        }
    memcpy(o, t, sizeof t);
}
/* Matrix3::determinant */
static double mat_det(const double m[9]) {
    return M(m,0,0) * (M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2))
         - M(m,1,0) * (M(m,0,1) * M(m,2,2) - M(m,2,1) * M(m,0,2))
         + M(m,2,0) * (M(m,0,1) * M(m,1,2) - M(m,1,1) * M(m,0,2));
}
/* Matrix3::invert = from_cols(c1 x c2 / det, c2 x c0 / det, c0 x c1 / det).transpose()
 * (Basis3::invert calls it and unwraps: src/baproblem.rs:162,174) */
static void mat_invert(const double m[9], double o[9]) {
    double det = mat_det(m);
    double a[3], b[3], c[3];
    cross3(&m[3], &m[6], a);
    cross3(&m[6], &m[0], b);
    cross3(&m[0], &m[3], c);
    for (int i = 0; i < 3; ++i) { a[i] /= det; b[i] /= det; c[i] /= det; }
    /* columns a,b,c then transpose: out[c][r] = col_r[c] */
    for (int i = 0; i < 3; ++i) {
        M(o, i, 0) = a[i];
        M(o, i, 1) = b[i];
        M(o, i, 2) = c[i];
    }
}
/* Matrix3::from_axis_angle (Basis3::from_axis_angle wraps it) */
static void mat_from_axis_angle(const double ax[3], double angle, double o[9]) {
    double s = sin(angle), c = cos(angle);
    double k = 1.0 - c;
    M(o,0,0) = k * ax[0] * ax[0] + c;
    M(o,0,1) = k * ax[0] * ax[1] + s * ax[2];
    M(o,0,2) = k * ax[0] * ax[2] - s * ax[1];
    M(o,1,0) = k * ax[0] * ax[1] - s * ax[2];
    M(o,1,1) = k * ax[1] * ax[1] + c;
    M(o,1,2) = k * ax[1] * ax[2] + s * ax[0];
    M(o,2,0) = k * ax[0] * ax[2] + s * ax[1];
    M(o,2,1) = k * ax[1] * ax[2] - s * ax[0];
    M(o,2,2) = k * ax[2] * ax[2] + c;
}
/* Matrix3::from_angle_x / from_angle_y (Matrix3::new is column-major) */
static void mat_from_angle_x(double th, double o[9]) {
    double s = sin(th), c = cos(th);
    double t[9] = { 1, 0, 0,  0, c, s,  0, -s, c };
    memcpy(o, t, sizeof t);
}
static void mat_from_angle_y(double th, double o[9]) {
    double s = sin(th), c = cos(th);
    double t[9] = { c, 0, -s,  0, 1, 0,  s, 0, c };
    memcpy(o, t, sizeof t);
}
/* From<Matrix3> for Quaternion (trace method), q = {s, x, y, z} */
static void quat_from_mat(const double m[9], double q[4]) {
    double trace = M(m,0,0) + M(m,1,1) + M(m,2,2);
    double w, x, y, z, s;
    if (trace >= 0.0) {
        s = sqrt(1.0 + trace);
        w = 0.5 * s; s = 0.5 / s;
        x = (M(m,1,2) - M(m,2,1)) * s;
        y = (M(m,2,0) - M(m,0,2)) * s;
        z = (M(m,0,1) - M(m,1,0)) * s;
    } else if (M(m,0,0) > M(m,1,1) && M(m,0,0) > M(m,2,2)) {
        s = sqrt((M(m,0,0) - M(m,1,1) - M(m,2,2)) + 1.0);
        x = 0.5 * s; s = 0.5 / s;
        y = (M(m,1,0) + M(m,0,1)) * s;
        z = (M(m,0,2) + M(m,2,0)) * s;
        w = (M(m,1,2) - M(m,2,1)) * s;
    } else if (M(m,1,1) > M(m,2,2)) {
        s = sqrt((M(m,1,1) - M(m,0,0) - M(m,2,2)) + 1.0);
        y = 0.5 * s; s = 0.5 / s;
        z = (M(m,2,1) + M(m,1,2)) * s;
        x = (M(m,1,0) + M(m,0,1)) * s;
        w = (M(m,2,0) - M(m,0,2)) * s;
    } else {
        s = sqrt((M(m,2,2) - M(m,0,0) - M(m,1,1)) + 1.0);
        z = 0.5 * s; s = 0.5 / s;
        x = (M(m,0,2) + M(m,2,0)) * s;
        y = (M(m,2,1) + M(m,1,2)) * s;
        w = (M(m,0,1) - M(m,1,0)) * s;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}
/* From<Quaternion> for Matrix3 */
static void mat_from_quat(const double q[4], double o[9]) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    double x2 = x + x, y2 = y + y, z2 = z + z;
    double xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
    double yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    double sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    double t[9] = {
        1.0 - yy2 - zz2, xy2 + sz2, xz2 - sy2,
        xy2 - sz2, 1.0 - xx2 - zz2, yz2 + sx2,
        xz2 + sy2, yz2 - sx2, 1.0 - xx2 - yy2,
    };
    memcpy(o, t, sizeof t);
}

/* ------------------------------------------------------------------ */
/* Rodrigues <-> rotation: src/baproblem.rs:78-102                     */
/* ------------------------------------------------------------------ */

/* from_rodrigues, src/baproblem.rs:78-90 */
void orc_from_rodrigues(const double w[3], double R[9]) {
    double theta2 = dot3(w, w);
    if (theta2 > DBL_EPSILON) {                 /* Rad::<f64>::default_epsilon() */
        double angle = mag3(w);
        double axis[3];
        normalize3(w, axis);
        mat_from_axis_angle(axis, angle, R);
    } else {
        /* Matrix3::new(1, x2, -x1, -x2, 1, x0, x1, -x0, 1), column-major */
        double m[9] = { 1.0, w[2], -w[1],  -w[2], 1.0, w[0],  w[1], -w[0], 1.0 };
        double q[4];
        quat_from_mat(m, q);
        mat_from_quat(q, R);
    }
}

/* to_rodrigues, src/baproblem.rs:93-102 */
void orc_to_rodrigues(const double R[9], double w[3]) {
    double q[4];
    quat_from_mat(R, q);
    double angle = 2.0 * acos(q[0]);
    if ((1.0 - q[0] * q[0]) < DBL_EPSILON) {
        w[0] = w[1] = w[2] = 0.0;
    } else {
        double d = sqrt(1.0 - q[0] * q[0]);
        double axis[3] = { q[1] / d, q[2] / d, q[3] / d };
        double n[3];
        normalize3(axis, n);
        w[0] = n[0] * angle; w[1] = n[1] * angle; w[2] = n[2] * angle;
    }
}

/* SnavelyCamera::from_vec / to_vec, src/baproblem.rs:180-202 */
void orc_camera_from_bal(const double bal9[9], double cam15[15]) {
    orc_from_rodrigues(bal9, cam15);
    for (int i = 0; i < 6; ++i) cam15[9 + i] = bal9[3 + i];
}
void orc_camera_to_bal(const double cam15[15], double bal9[9]) {
    orc_to_rodrigues(cam15, bal9);
    for (int i = 0; i < 6; ++i) bal9[3 + i] = cam15[9 + i];
}

/* ------------------------------------------------------------------ */
/* Camera trait for SnavelyCamera: src/baproblem.rs:140-176            */
/* ------------------------------------------------------------------ */

/* project_world, :141-143   dir.rotate_point(p) + loc */
void orc_project_world(const double cam[15], const double p[3], double q[3]) {
    double rp[3];
    mat_vec(cam, p, rp);
    q[0] = rp[0] + cam[9]; q[1] = rp[1] + cam[10]; q[2] = rp[2] + cam[11];
}

/* p.magnitude().powf(4.0), :149.  Rust's f64::powf is the platform libm's pow (glibc here, as on the machine the
 * reference would run on), and that is ALL the oracle evaluates: one arithmetic, the reference's.  glibc's pow is
 * accurate to ~0.52 ulp, NOT correctly rounded: pow(sqrt(n), 4.0) differs from the correctly rounded fl(sqrt(n))^4 in
 * 0.09 % of draws, always by one ulp (tests/test_pow4.py measures it).  Rounds 1-5 had the device evaluate the correctly
 * rounded value and gave this file a second mode to match it; since round 6 the device runs a restatement of glibc's own
 * pow (city2ba_amd/csrc/pow4_libm.hpp, bit for bit this libm on every argument class) and the second mode is gone. */

/* x^4 correctly rounded -- NOT used by any projection here; kept so that tests/test_pow4.py can state how often and by
 * how much libm's pow differs from it (x^2 = h + l exactly, h^2 = hh + hl exactly, x^4 = hh + (hl + 2 h l + l^2)). */
double orc_pow4_cr(double x) {
    const double h = x * x;
    const double hh = h * h;
    if (!(hh < INFINITY) || hh == 0.0) return hh;  /* overflow, NaN, or underflow to zero: same class as pow() */
    const double l = fma(x, x, -h);
    const double hl = fma(h, h, -hh);
    double t = fma(2.0 * h, l, hl);
    t = fma(l, l, t);
    return hh + t;
}

/* x[i]^4 by libm's pow (what the projection below calls) and correctly rounded, for an array */
void orc_pow4_both(const double *x, int64_t n, double *out_libm, double *out_cr) {
    volatile double four = 4.0;                    /* the library call, not a compile-time expansion */
    for (int64_t i = 0; i < n; ++i) { out_libm[i] = pow(x[i], four); out_cr[i] = orc_pow4_cr(x[i]); }
}

static double magnitude_pow4(double mag2) {
    volatile double four = 4.0;
    return pow(sqrt(mag2), four);
}

/* project, :145-151 */
void orc_project(const double cam[15], const double q[3], double uv[2]) {
    double px = -q[0] / q[2], py = -q[1] / q[2];
    double mag2 = px * px + py * py;               /* Vector2::magnitude2 */
    double r = 1.0 + cam[13] * mag2 + cam[14] * magnitude_pow4(mag2);
    double fr = cam[12] * r;                       /* focal_length() * r */
    uv[0] = fr * px; uv[1] = fr * py;
}

/* center, :161-163   -(dir.invert().rotate_vector(loc)) */
void orc_center(const double cam[15], double c[3]) {
    double inv[9], v[3];
    mat_invert(cam, inv);
    mat_vec(inv, &cam[9], v);
    c[0] = -v[0]; c[1] = -v[1]; c[2] = -v[2];
}

/* to_world, :173-175   dir.invert().rotate_point(p - loc) */
void orc_to_world(const double cam[15], const double p[3], double o[3]) {
    double inv[9];
    double d[3] = { p[0] - cam[9], p[1] - cam[10], p[2] - cam[11] };
    mat_invert(cam, inv);
    mat_vec(inv, d, o);
}

/* from_position_direction, :153-159 */
void orc_from_position_direction(const double pos[3], const double R[9], double cam[15]) {
    double v[3];
    mat_vec(R, pos, v);
    memcpy(cam, R, 9 * sizeof(double));
    cam[9] = -1.0 * v[0]; cam[10] = -1.0 * v[1]; cam[11] = -1.0 * v[2];
    cam[12] = 1.0; cam[13] = 0.0; cam[14] = 0.0;
}

/* transform, :165-171.  NOTE the reference uses the OLD dir for the new loc. */
void orc_transform(const double cam[15], const double dR[9], const double dloc[3], double out[15]) {
    double c[3], s[3], v[3], nr[9];
    orc_center(cam, c);
    s[0] = c[0] + dloc[0]; s[1] = c[1] + dloc[1]; s[2] = c[2] + dloc[2];
    mat_vec(cam, s, v);
    mat_mul(cam, dR, nr);
    memcpy(out, nr, sizeof nr);
    out[9] = -1.0 * v[0]; out[10] = -1.0 * v[1]; out[11] = -1.0 * v[2];
    out[12] = cam[12]; out[13] = cam[13]; out[14] = cam[14];
}

/* Orientation helpers used by the synthetic generators, src/synthetic.rs:191-205,330.
 * Deg -> Rad is deg * (PI / 180) in cgmath. */
void orc_basis_from_angle_y_deg(double deg, double R[9]) {
    mat_from_angle_y(deg * (3.14159265358979323846 / 180.0), R);
}
void orc_basis_from_angle_x_rad(double rad, double R[9]) { mat_from_angle_x(rad, R); }
void orc_basis_from_axis_angle(const double ax[3], double angle, double R[9]) {
    mat_from_axis_angle(ax, angle, R);
}

/* ------------------------------------------------------------------ */
/* BAProblem batch methods over the CSR form of vis_graph              */
/*   row_ptr[n_cam+1], pt_idx[n_obs] (u64 = usize), uv[2*n_obs]        */
/*   (Vec<Vec<(usize,(f64,f64))>>, src/baproblem.rs:256-260)           */
/* ------------------------------------------------------------------ */

/* project every observation: the per-observation body of :272 */
void orc_project_observations(const double *cams15, int64_t n_cam, const double *pts,
                              const uint64_t *row_ptr, const uint64_t *pt_idx, double *uv_out) {
    for (int64_t c = 0; c < n_cam; ++c)
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o) {
            double q[3];
            orc_project_world(&cams15[15 * c], &pts[3 * pt_idx[o]], q);
            orc_project(&cams15[15 * c], q, &uv_out[2 * o]);
        }
}

/* total_reprojection_error, src/baproblem.rs:265-279 (sequential, camera-major) */
double orc_total_reprojection_error(const double *cams15, int64_t n_cam, const double *pts,
                                    const uint64_t *row_ptr, const uint64_t *pt_idx,
                                    const double *uv, double norm) {
    double total = 0.0;
    for (int64_t c = 0; c < n_cam; ++c) {
        double cam_sum = 0.0;
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o) {
            double q[3], p[2];
            orc_project_world(&cams15[15 * c], &pts[3 * pt_idx[o]], q);
            orc_project(&cams15[15 * c], q, p);
            cam_sum += pow(fabs(p[0] - uv[2 * o]), norm) + pow(fabs(p[1] - uv[2 * o + 1]), norm);
        }
        total += cam_sum;
    }
    return pow(total, 1.0 / norm);
}

/* Same sum WITHOUT the final ^(1/norm): what one GPU shard contributes before
 * the all-reduce (build-side helper; same loop as above). */
double orc_reprojection_error_sum(const double *cams15, int64_t n_cam, const double *pts,
                                  const uint64_t *row_ptr, const uint64_t *pt_idx,
                                  const double *uv, double norm) {
    double total = 0.0;
    for (int64_t c = 0; c < n_cam; ++c) {
        double cam_sum = 0.0;
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o) {
            double q[3], p[2];
            orc_project_world(&cams15[15 * c], &pts[3 * pt_idx[o]], q);
            orc_project(&cams15[15 * c], q, p);
            cam_sum += pow(fabs(p[0] - uv[2 * o]), norm) + pow(fabs(p[1] - uv[2 * o + 1]), norm);
        }
        total += cam_sum;
    }
    return total;
}

/* mean, :282-289  fold(0, a + b/num) over centers then points */
void orc_mean(const double *cams15, int64_t n_cam, const double *pts, int64_t n_pts, double out[3]) {
    double num = (double)(n_cam + n_pts);
    double a[3] = { 0, 0, 0 };
    for (int64_t c = 0; c < n_cam; ++c) {
        double ctr[3];
        orc_center(&cams15[15 * c], ctr);
        for (int k = 0; k < 3; ++k) a[k] = a[k] + ctr[k] / num;
    }
    for (int64_t p = 0; p < n_pts; ++p)
        for (int k = 0; k < 3; ++k) a[k] = a[k] + pts[3 * p + k] / num;
    memcpy(out, a, sizeof a);
}

/* std, :292-304 */
void orc_std(const double *cams15, int64_t n_cam, const double *pts, int64_t n_pts, double out[3]) {
    double num = (double)(n_cam + n_pts);
    double mean[3], a[3] = { 0, 0, 0 };
    orc_mean(cams15, n_cam, pts, n_pts, mean);
    for (int64_t c = 0; c < n_cam; ++c) {
        double ctr[3];
        orc_center(&cams15[15 * c], ctr);
        for (int k = 0; k < 3; ++k) a[k] = a[k] + (ctr[k] - mean[k]) * (ctr[k] - mean[k]);
    }
    for (int64_t p = 0; p < n_pts; ++p)
        for (int k = 0; k < 3; ++k)
            a[k] = a[k] + (pts[3 * p + k] - mean[k]) * (pts[3 * p + k] - mean[k]);
    for (int k = 0; k < 3; ++k) out[k] = sqrt(a[k] / num);
}

/* extent, :307-331 (f64::min / f64::max ignore NaN like fmin/fmax) */
void orc_extent(const double *cams15, int64_t n_cam, const double *pts, int64_t n_pts,
                double mn[3], double mx[3]) {
    for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
    for (int64_t c = 0; c < n_cam; ++c) {
        double ctr[3];
        orc_center(&cams15[15 * c], ctr);
        for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], ctr[k]); mx[k] = fmax(mx[k], ctr[k]); }
    }
    for (int64_t p = 0; p < n_pts; ++p)
        for (int k = 0; k < 3; ++k) {
            mn[k] = fmin(mn[k], pts[3 * p + k]); mx[k] = fmax(mx[k], pts[3 * p + k]);
        }
}

/* dimensions, :334-337 */
void orc_dimensions(const double *cams15, int64_t n_cam, const double *pts, int64_t n_pts, double d[3]) {
    double mn[3], mx[3];
    orc_extent(cams15, n_cam, pts, n_pts, mn, mx);
    for (int k = 0; k < 3; ++k) d[k] = mx[k] - mn[k];
}

/* ------------------------------------------------------------------ */
/* Visibility predicate shared by the three generator loops:           */
/* src/synthetic.rs:285-291, :368-375; src/generate.rs:448-454.        */
/* keep = |center - p| < max_dist && q.z <= 0 && -1<=u<=1 && -1<=v<=1  */
/* ------------------------------------------------------------------ */
void orc_visibility_pairs(const double *cams15, const double *pts, const uint32_t *cam_idx,
                          const uint32_t *pt_idx, int64_t n_pairs, double max_dist,
                          double *uv_out, uint8_t *keep) {
    for (int64_t i = 0; i < n_pairs; ++i) {
        const double *cam = &cams15[15 * (int64_t)cam_idx[i]];
        const double *p = &pts[3 * (int64_t)pt_idx[i]];
        double q[3], c[3], d[3], uv[2] = { NAN, NAN };
        uint8_t k = 0;
        orc_project_world(cam, p, q);
        orc_center(cam, c);
        d[0] = c[0] - p[0]; d[1] = c[1] - p[1]; d[2] = c[2] - p[2];
        if (mag3(d) < max_dist && q[2] <= 0.0) {
            orc_project(cam, q, uv);
            if (uv[0] >= -1.0 && uv[0] <= 1.0 && uv[1] >= -1.0 && uv[1] <= 1.0) k = 1;
        }
        uv_out[2 * i] = uv[0]; uv_out[2 * i + 1] = uv[1];
        keep[i] = k;
    }
}

/* Occlusion rays of generate::visibility_graph, src/generate.rs:455-476:        */
/* ray = (center as f32, normalize(point - center) as f32), tfar = |dir| as f32   */
/* - 1e-6; the pair is kept iff the scene does not occlude the ray.  The scene    */
/* query itself is Embree's (embree-rs 0.3 / libembree3, neither vendored nor     */
/* installed here: parity unpinned); restated as the textbook Moeller-Trumbore    */
/* test in float32 against every triangle, occluded iff some 0 < t <= tfar.       */
void orc_occlusion_filter(const double *cams15, const double *pts, const uint32_t *cam_idx,
                          const uint32_t *pt_idx, int64_t n_pairs, const float *tri9, int64_t n_tri,
                          uint8_t *keep) {
    for (int64_t i = 0; i < n_pairs; ++i) {
        const double *p = &pts[3 * (int64_t)pt_idx[i]];
        double c[3], e[3];
        orc_center(&cams15[15 * (int64_t)cam_idx[i]], c);
        e[0] = p[0] - c[0]; e[1] = p[1] - c[1]; e[2] = p[2] - c[2];
        const double mag = mag3(e), inv = 1.0 / mag;
        const float o[3] = { (float)c[0], (float)c[1], (float)c[2] };
        const float d[3] = { (float)(e[0] * inv), (float)(e[1] * inv), (float)(e[2] * inv) };
        const float tfar = (float)mag - 1e-6f;
        uint8_t occluded = 0;
        for (int64_t t = 0; t < n_tri && !occluded; ++t) {
            const float *q = &tri9[9 * t];
            const float e1[3] = { q[3] - q[0], q[4] - q[1], q[5] - q[2] };
            const float e2[3] = { q[6] - q[0], q[7] - q[1], q[8] - q[2] };
            const float pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
            const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
            if (det == 0.0f) continue;
            const float idet = 1.0f / det;
            const float tv[3] = { o[0] - q[0], o[1] - q[1], o[2] - q[2] };
            const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * idet;
            if (u < 0.0f || u > 1.0f) continue;
            const float qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
            const float w = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * idet;
            if (w < 0.0f || u + w > 1.0f) continue;
            const float th = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * idet;
            if (th > 0.0f && th <= tfar) occluded = 1;
        }
        keep[i] = occluded ? 0 : 1;
    }
}

/* ------------------------------------------------------------------ */
/* Residual + 2x(9+3) Jacobian.  NOT IN THE REFERENCE (parity unpinned */
/* by it).  r = project(project_world(X)) - uv_obs, matching the sign  */
/* of src/baproblem.rs:273; columns follow to_vec order (:189-202):    */
/* Jc row-major 2x9 [w0 w1 w2 t0 t1 t2 f k1 k2], Jp row-major 2x3.     */
/* d(R X)/dw uses the Gallego-Yezzi closed form                        */
/*   -R [X]x (w w^T + (R^T - I)[w]x) / |w|^2 ,   small angle: -[X]x    */
/* (a deliberately different algebra from the GPU kernel's left-       */
/* Jacobian form, so the two check each other; both are checked        */
/* against mpmath in tests/golden/).                                   */
/* ------------------------------------------------------------------ */
static void skew(const double v[3], double S[3][3]) {
    S[0][0] = 0;     S[0][1] = -v[2]; S[0][2] = v[1];
    S[1][0] = v[2];  S[1][1] = 0;     S[1][2] = -v[0];
    S[2][0] = -v[1]; S[2][1] = v[0];  S[2][2] = 0;
}
static void mm33(const double A[3][3], const double B[3][3], double C[3][3]) {
    double T[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
    memcpy(C, T, sizeof T);
}

/* Core: R (cam15) and the Rodrigues vector w the columns refer to are passed
 * explicitly so that the derivative is taken w.r.t. exactly the parameters the
 * caller holds.  "state mode": w = to_rodrigues(R) (what to_vec writes);
 * "bal mode": w is the 9-vector's own w and R = from_rodrigues(w). */
void orc_residual_jacobian_one(const double cam[15], const double w[3], const double X[3],
                               const double uv_obs[2], double r[2], double Jc[18], double Jp[6]) {
    double Rm[3][3];                       /* row/col math matrix */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rm[i][j] = M(cam, j, i);
    double q[3], uv[2];
    orc_project_world(cam, X, q);
    orc_project(cam, q, uv);
    r[0] = uv[0] - uv_obs[0]; r[1] = uv[1] - uv_obs[1];

    double f = cam[12], k1 = cam[13], k2 = cam[14];
    double px = -q[0] / q[2], py = -q[1] / q[2];
    double n = px * px + py * py;
    double rad = 1.0 + k1 * n + k2 * n * n;
    double c = 2.0 * k1 + 4.0 * k2 * n;
    /* duv/dp = f (rad I + c p p^T) */
    double B[2][2] = { { f * (rad + c * px * px), f * c * px * py },
                       { f * c * px * py, f * (rad + c * py * py) } };
    /* dp/dq */
    double z = q[2];
    double P[2][3] = { { -1.0 / z, 0.0, q[0] / (z * z) }, { 0.0, -1.0 / z, q[1] / (z * z) } };
    double A[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = B[i][0] * P[0][j] + B[i][1] * P[1][j];

    /* D = d(R X)/dw */
    double D[3][3];
    double th2 = dot3(w, w);
    double Sw[3][3];
    skew(w, Sw);
    if (th2 > 1e-8) {
        /* Gallego-Yezzi; conditioning ~ eps/|w|, fine above |w| = 1e-4 */
        double SX[3][3], T[3][3], G[3][3], RtI[3][3];
        skew(X, SX);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) RtI[i][j] = Rm[j][i] - (i == j ? 1.0 : 0.0);
        mm33(RtI, Sw, T);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) G[i][j] = (w[i] * w[j] + T[i][j]) / th2;
        mm33(SX, G, T);
        mm33(Rm, T, D);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i][j] = -D[i][j];
    } else {
        /* |w| <= 1e-4: series  -[R X]x (I + [w]x/2 + [w]x^2/6), truncation O(|w|^3) */
        double y[3] = { q[0] - cam[9], q[1] - cam[10], q[2] - cam[11] };
        double Sy[3][3], K2[3][3], Jl[3][3];
        skew(y, Sy);
        mm33(Sw, Sw, K2);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Jl[i][j] = (i == j ? 1.0 : 0.0) + 0.5 * Sw[i][j] + K2[i][j] / 6.0;
        mm33(Sy, Jl, D);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i][j] = -D[i][j];
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 3; ++j) {
            Jc[9 * i + j] = A[i][0] * D[0][j] + A[i][1] * D[1][j] + A[i][2] * D[2][j];
            Jc[9 * i + 3 + j] = A[i][j];
            Jp[3 * i + j] = A[i][0] * Rm[0][j] + A[i][1] * Rm[1][j] + A[i][2] * Rm[2][j];
        }
    }
    Jc[6] = rad * px;          Jc[15] = rad * py;
    Jc[7] = f * n * px;        Jc[16] = f * n * py;
    Jc[8] = f * n * n * px;    Jc[17] = f * n * n * py;
}

/* state mode over the CSR graph: columns refer to w = to_rodrigues(R) */
void orc_residual_jacobian(const double *cams15, int64_t n_cam, const double *pts,
                           const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv,
                           double *r, double *Jc, double *Jp) {
    for (int64_t c = 0; c < n_cam; ++c) {
        double w[3];
        orc_to_rodrigues(&cams15[15 * c], w);
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o)
            orc_residual_jacobian_one(&cams15[15 * c], w, &pts[3 * pt_idx[o]], &uv[2 * o],
                                      &r[2 * o], &Jc[18 * o], &Jp[6 * o]);
    }
}

/* bal mode over the CSR graph: cameras given as 9-vectors; R = from_rodrigues(w) */
void orc_residual_jacobian_bal(const double *bal9, int64_t n_cam, const double *pts,
                               const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv,
                               double *r, double *Jc, double *Jp) {
    for (int64_t c = 0; c < n_cam; ++c) {
        double cam[15];
        orc_camera_from_bal(&bal9[9 * c], cam);
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o)
            orc_residual_jacobian_one(cam, &bal9[9 * c], &pts[3 * pt_idx[o]], &uv[2 * o],
                                      &r[2 * o], &Jc[18 * o], &Jp[6 * o]);
    }
}

/* ------------------------------------------------------------------ */
/* Noise.  The reference draws from rand::thread_rng() (unseeded,      */
/* src/noise.rs:38-40,91,96,135,149,160-163): bit parity is impossible */
/* by construction.  The build defines a counter-based generator       */
/* (Philox4x32-10, Salmon et al. SC'11) keyed by (seed; stream, entity,*/
/* slot) so results do not depend on sharding; normals by Box-Muller.  */
/* Everything around the draws restates the reference's algebra.       */
/* ------------------------------------------------------------------ */
enum { ORC_STREAM_DRIFT_CAM = 1, ORC_STREAM_DRIFT_PT = 2, ORC_STREAM_NOISE_CAM = 3,
       ORC_STREAM_NOISE_PT = 4, ORC_STREAM_NOISE_OBS = 5 };

void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Philox2x32-10 (Random123's philox2x32_R(10, ...)): the observation stream's generator */
void orc_philox2x32_10(const uint32_t ctr_in[2], uint32_t key, uint32_t out[2]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], k = key;
    for (int i = 0; i < 10; ++i) {
        uint64_t p = (uint64_t)0xD256D193u * c0;
        uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}

/* two independent N(0,1) for (seed, stream, entity, slot) */
void orc_normal_pair(uint64_t seed, uint32_t stream, uint64_t entity, uint32_t slot, double z[2]) {
    uint32_t ctr[4] = { (uint32_t)entity, (uint32_t)(entity >> 32), slot, stream };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    uint64_t a = ((uint64_t)o[1] << 32) | o[0];
    uint64_t b = ((uint64_t)o[3] << 32) | o[2];
    double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;    /* (0,1] */
    double u2 = (double)(b >> 11) * 0x1.0p-53;          /* [0,1) */
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    z[0] = rad * cos(ang);
    z[1] = rad * sin(ang);
}

/* closest-to-origin element of centers ++ points, src/noise.rs:75-87:
 * fold1(|x,y| if |x| < |y| {x} else {y}) => strict <, ties go to the LATER element */
void orc_drift_origin(const double *cams15, int64_t n_cam, const double *pts, int64_t n_pts,
                      double origin[3], int64_t *index) {
    double best[3] = { 0, 0, 0 }, bd = 0;
    int64_t bi = -1;
    for (int64_t i = 0; i < n_cam + n_pts; ++i) {
        double x[3];
        if (i < n_cam) orc_center(&cams15[15 * i], x);
        else memcpy(x, &pts[3 * (i - n_cam)], sizeof x);
        double d = mag3(x);
        if (bi < 0 || !(bd < d)) { memcpy(best, x, sizeof x); bd = d; bi = i; }
    }
    memcpy(origin, best, sizeof best);
    if (index) *index = bi;
}

/* add_drift, src/noise.rs:68-116, in place on cams15 / pts */
void orc_add_drift(double *cams15, int64_t n_cam, double *pts, int64_t n_pts,
                   double strength, double angle_strength, double std, const double dir[3],
                   uint64_t seed) {
    double origin[3];
    orc_drift_origin(cams15, n_cam, pts, n_pts, origin, NULL);
    for (int64_t i = 0; i < n_cam; ++i) {
        double *cam = &cams15[15 * i];
        double c[3], d[3], z[2], dR[9], dl[3], out[15];
        orc_center(cam, c);
        d[0] = c[0] - origin[0]; d[1] = c[1] - origin[1]; d[2] = c[2] - origin[2];
        double distance = mag3(d);
        orc_normal_pair(seed, ORC_STREAM_DRIFT_CAM, (uint64_t)i, 0, z);
        double va = 1.0 + std * z[0];           /* angle draw comes first (:104-107) */
        double vt = 1.0 + std * z[1];
        double angle = angle_strength * va * pow(distance, 1.2);
        for (int k = 0; k < 3; ++k) dl[k] = dir[k] * strength * vt * distance * distance;
        mat_from_angle_x(angle, dR);
        orc_transform(cam, dR, dl, out);
        memcpy(cam, out, sizeof out);
    }
    for (int64_t j = 0; j < n_pts; ++j) {
        double *p = &pts[3 * j];
        double d[3] = { p[0] - origin[0], p[1] - origin[1], p[2] - origin[2] };
        double distance = mag3(d), z[2];
        orc_normal_pair(seed, ORC_STREAM_DRIFT_PT, (uint64_t)j, 0, z);
        double v = 1.0 + std * z[0];
        for (int k = 0; k < 3; ++k) p[k] = p[k] + dir[k] * strength * v * distance * distance;
    }
}

/* add_drift_normalized, src/noise.rs:47-56 */
void orc_add_drift_normalized(double *cams15, int64_t n_cam, double *pts, int64_t n_pts,
                              double strength, double angle_strength, double std, uint64_t seed) {
    double s[3], dir[3];
    orc_std(cams15, n_cam, pts, n_pts, s);
    normalize3(s, dir);
    double bal_std = mag3(s);
    orc_add_drift(cams15, n_cam, pts, n_pts, strength * bal_std, angle_strength, std, dir, seed);
}

/* add_noise, src/noise.rs:119-177.  uv is the CSR-ordered observation array;
 * obs_offset = global index of uv[0] (for sharded use). */
void orc_add_noise(double *cams15, int64_t n_cam, double *pts, int64_t n_pts, double *uv,
                   int64_t n_obs, uint64_t obs_offset, double translation_std,
                   double rotation_std, double point_std, double observations_std, uint64_t seed) {
    double s[3];
    orc_std(cams15, n_cam, pts, n_pts, s);
    double bal_std = mag3(s);
    for (int64_t i = 0; i < n_cam; ++i) {
        double *cam = &cams15[15 * i];
        double z0[2], z1[2], z2[2], z3[2];
        orc_normal_pair(seed, ORC_STREAM_NOISE_CAM, (uint64_t)i, 0, z0);
        orc_normal_pair(seed, ORC_STREAM_NOISE_CAM, (uint64_t)i, 1, z1);
        orc_normal_pair(seed, ORC_STREAM_NOISE_CAM, (uint64_t)i, 2, z2);
        orc_normal_pair(seed, ORC_STREAM_NOISE_CAM, (uint64_t)i, 3, z3);
        double a[3] = { z0[0], z0[1], z1[0] }, ax[3], b[3] = { z2[0], z2[1], z3[0] }, bx[3];
        normalize3(a, ax);
        double ang = 0.0 + rotation_std * z1[1];
        normalize3(b, bx);
        double tr = 0.0 + translation_std * z3[1];
        double dR[9], dl[3], out[15];
        mat_from_axis_angle(ax, ang, dR);
        for (int k = 0; k < 3; ++k) dl[k] = bx[k] * bal_std * tr;
        orc_transform(cam, dR, dl, out);
        memcpy(cam, out, sizeof out);
    }
    for (int64_t j = 0; j < n_pts; ++j) {
        double z0[2], z1[2];
        orc_normal_pair(seed, ORC_STREAM_NOISE_PT, (uint64_t)j, 0, z0);
        orc_normal_pair(seed, ORC_STREAM_NOISE_PT, (uint64_t)j, 1, z1);
        double a[3] = { z0[0], z0[1], z1[0] }, ax[3];
        normalize3(a, ax);
        double m = 0.0 + point_std * z1[1];
        for (int k = 0; k < 3; ++k) pts[3 * j + k] = pts[3 * j + k] + ax[k] * m;
    }
    for (int64_t o = 0; o < n_obs; ++o) {
        /* ONE Philox2x32-10 block per observation: counter = (low word of the global observation index, its high word
         * xor the seed's high word), key = the seed's low word.  Word 0 -> the radius uniform of the magnitude draw,
         * (w0 + 1) 2^-32 in (0, 1]; word 1: high half -> its angle, low half -> the direction (16-bit fractions of a
         * turn).  unit_random::<Vector2>() normalises a pair of standard normals (src/noise.rs:35-45): its direction is
         * uniform on the circle and its radius cancels, so the pair is drawn with radius 1, i.e. as (cos, sin) of a
         * uniform angle, and normalised like the reference does; the magnitude is Normal(0, std) by Box-Muller
         * (src/noise.rs:160-163). */
        uint64_t ent = obs_offset + (uint64_t)o;
        uint32_t ctr[2] = { (uint32_t)ent, (uint32_t)(ent >> 32) ^ (uint32_t)(seed >> 32) };
        uint32_t w[2];
        orc_philox2x32_10(ctr, (uint32_t)seed, w);
        double u1 = ((double)w[0] + 1.0) * 0x1.0p-32;                         /* (0,1] */
        double dir = 6.283185307179586476925286766559 * ((double)(w[1] & 0xffffu) * 0x1.0p-16);
        double ang = 6.283185307179586476925286766559 * ((double)(w[1] >> 16) * 0x1.0p-16);
        double z = sqrt(-2.0 * log(u1)) * cos(ang);
        double nx = cos(dir), ny = sin(dir);
        double m = sqrt(nx * nx + ny * ny);       /* powf(2.0) == x*x exactly */
        double r = 0.0 + observations_std * z;
        uv[2 * o] = uv[2 * o] + nx / m * r;
        uv[2 * o + 1] = uv[2 * o + 1] + ny / m * r;
    }
}

/* add_sin_noise, src/noise.rs:388-416 */
void orc_add_sin_noise(double *cams15, int64_t n_cam, double *pts, int64_t n_pts,
                       const double dir[3], const double noise_dir[3], double strength,
                       double frequency) {
    double dim[3], nd[3];
    orc_dimensions(cams15, n_cam, pts, n_pts, dim);
    for (int k = 0; k < 3; ++k) if (dim[k] == 0.0) dim[k] = 1e-8;
    normalize3(noise_dir, nd);
    double identity[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int64_t i = 0; i < n_cam; ++i) {
        double *cam = &cams15[15 * i];
        double c[3], e[3], dl[3], out[15];
        orc_center(cam, c);
        for (int k = 0; k < 3; ++k) e[k] = c[k] / dim[k];
        double s = sin(dot3(e, dir) * frequency * 3.14159265358979323846) * strength;
        for (int k = 0; k < 3; ++k) dl[k] = nd[k] * s;      /* scalar * vector */
        orc_transform(cam, identity, dl, out);
        memcpy(cam, out, sizeof out);
    }
    for (int64_t j = 0; j < n_pts; ++j) {
        double *p = &pts[3 * j], e[3];
        for (int k = 0; k < 3; ++k) e[k] = p[k] / dim[k];
        double s = sin(dot3(e, dir) * frequency * 3.14159265358979323846) * strength;
        for (int k = 0; k < 3; ++k) p[k] = p[k] + nd[k] * s;
    }
}

/* ------------------------------------------------------------------ */
/* Faithful-layout CPU baseline leg (bench.py cpu_baseline, kind       */
/* "port"): project + Jacobian + error over the CSR graph on one       */
/* thread, camera-major like src/baproblem.rs:266-278.                 */
/* ------------------------------------------------------------------ */
double orc_bench_residual_jacobian(const double *cams15, int64_t n_cam, const double *pts,
                                   const uint64_t *row_ptr, const uint64_t *pt_idx,
                                   const double *uv, double *r, double *Jc, double *Jp) {
    double total = 0.0;
    for (int64_t c = 0; c < n_cam; ++c) {
        double cam_sum = 0.0;
        double w[3];
        orc_to_rodrigues(&cams15[15 * c], w);
        for (uint64_t o = row_ptr[c]; o < row_ptr[c + 1]; ++o) {
            orc_residual_jacobian_one(&cams15[15 * c], w, &pts[3 * pt_idx[o]], &uv[2 * o],
                                      &r[2 * o], &Jc[18 * o], &Jp[6 * o]);
            cam_sum += r[2 * o] * r[2 * o] + r[2 * o + 1] * r[2 * o + 1];
        }
        total += cam_sum;
    }
    return total;
}

/* ------------------------------------------------------------------ */
/* CPU baselines of bench.py (cpu_baseline leg only).  Timed INSIDE C  */
/* so that no Python dispatch is in the numbers.                       */
/*                                                                     */
/* layout 0 "faithful": the reference's own storage -- vis_graph is    */
/*   Vec<Vec<(usize, (f64, f64))>> (src/baproblem.rs:256-260): one heap */
/*   allocation per camera holding 24-byte AoS records -- walked        */
/*   camera-major with sequential per-camera sums and                   */
/*   abs().powf(norm) per term like total_reprojection_error            */
/*   (:265-279).  The Jacobian has no reference implementation; it is   */
/*   the oracle's closed form in both layouts.                          */
/* layout 1 "optimised-CPU": flat CSR arrays, r*r instead of pow.       */
/* threads > 1: contiguous camera ranges of equal camera count, one     */
/*   pthread each -- the static split rayon's par_iter makes over       */
/*   cameras in the reference's visibility loops                        */
/*   (src/synthetic.rs:268-269, src/generate.rs:434-435).               */
/* ------------------------------------------------------------------ */
#include <pthread.h>
#include <time.h>

typedef struct { uint64_t idx; double u, v; } orc_obs_t;              /* (usize, (f64, f64)) */
typedef struct { orc_obs_t *data; size_t len; } orc_obsvec_t;         /* Vec<(usize, (f64, f64))> */
typedef struct { int64_t n_cam; orc_obsvec_t *rows; } orc_visgraph_t; /* Vec<Vec<..>> */

orc_visgraph_t *orc_visgraph_from_csr(int64_t n_cam, const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    orc_visgraph_t *g = (orc_visgraph_t *)malloc(sizeof *g);
    if (!g) return NULL;
    g->n_cam = n_cam;
    g->rows = (orc_obsvec_t *)calloc((size_t)(n_cam > 0 ? n_cam : 1), sizeof *g->rows);
    if (!g->rows) { free(g); return NULL; }
    for (int64_t c = 0; c < n_cam; ++c) {
        const size_t len = (size_t)(row_ptr[c + 1] - row_ptr[c]);
        g->rows[c].len = len;
        g->rows[c].data = (orc_obs_t *)malloc((len ? len : 1) * sizeof(orc_obs_t));   /* one allocation per camera */
        if (!g->rows[c].data) { g->rows[c].len = 0; continue; }
        for (size_t k = 0; k < len; ++k) {
            const uint64_t o = row_ptr[c] + k;
            g->rows[c].data[k].idx = pt_idx[o];
            g->rows[c].data[k].u = uv[2 * o];
            g->rows[c].data[k].v = uv[2 * o + 1];
        }
    }
    return g;
}

void orc_visgraph_free(orc_visgraph_t *g) {
    if (!g) return;
    for (int64_t c = 0; c < g->n_cam; ++c) free(g->rows[c].data);
    free(g->rows);
    free(g);
}

typedef struct {
    int layout;
    const double *cams15, *pts, *uv;
    const uint64_t *row_ptr, *pt_idx;
    const orc_visgraph_t *graph;
    double norm;
    double *r, *Jc, *Jp;
    int64_t c0, c1;
    double total;
} orc_job_t;

static void *orc_bench_range(void *arg) {
    orc_job_t *j = (orc_job_t *)arg;
    double total = 0.0;
    for (int64_t c = j->c0; c < j->c1; ++c) {
        const double *cam = &j->cams15[15 * c];
        double cam_sum = 0.0, w[3];
        orc_to_rodrigues(cam, w);
        const uint64_t o0 = j->row_ptr[c];
        if (j->layout == 0) {
            const orc_obsvec_t *row = &j->graph->rows[c];
            for (size_t k = 0; k < row->len; ++k) {
                const uint64_t o = o0 + k;
                const double ob[2] = { row->data[k].u, row->data[k].v };
                orc_residual_jacobian_one(cam, w, &j->pts[3 * row->data[k].idx], ob, &j->r[2 * o], &j->Jc[18 * o], &j->Jp[6 * o]);
                cam_sum += pow(fabs(j->r[2 * o]), j->norm) + pow(fabs(j->r[2 * o + 1]), j->norm);
            }
        } else {
            for (uint64_t o = o0; o < j->row_ptr[c + 1]; ++o) {
                orc_residual_jacobian_one(cam, w, &j->pts[3 * j->pt_idx[o]], &j->uv[2 * o], &j->r[2 * o], &j->Jc[18 * o], &j->Jp[6 * o]);
                cam_sum += j->r[2 * o] * j->r[2 * o] + j->r[2 * o + 1] * j->r[2 * o + 1];
            }
        }
        total += cam_sum;
    }
    j->total = total;
    return NULL;
}

/* Runs whole passes over cameras [0, n_cam) until `seconds` have elapsed (at least one, at most max_passes); returns
 * the number of passes, *elapsed the wall time of exactly those passes, *total the reduced error sum of the last one. */
int64_t orc_bench_run(int layout, int threads, double seconds, int64_t max_passes, const double *cams15, int64_t n_cam,
                      const double *pts, const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv,
                      const orc_visgraph_t *graph, double norm, double *r, double *Jc, double *Jp, double *elapsed,
                      double *total) {
    if (threads < 1) threads = 1;
    if (threads > n_cam && n_cam > 0) threads = (int)n_cam;
    orc_job_t *jobs = (orc_job_t *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof *tids);
    if (!jobs || !tids) { free(jobs); free(tids); return -1; }
    for (int t = 0; t < threads; ++t) {
        orc_job_t j = { layout, cams15, pts, uv, row_ptr, pt_idx, graph, norm, r, Jc, Jp,
                        n_cam * t / threads, n_cam * (t + 1) / threads, 0.0 };
        jobs[t] = j;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int64_t passes = 0;
    double el = 0.0, sum = 0.0;
    do {
        if (threads == 1) {
            orc_bench_range(&jobs[0]);
        } else {
            for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, orc_bench_range, &jobs[t]);
            for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
        }
        sum = 0.0;
        for (int t = 0; t < threads; ++t) sum += jobs[t].total;
        ++passes;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    } while (el < seconds && passes < max_passes);
    *elapsed = el;
    *total = sum;
    free(jobs);
    free(tids);
    return passes;
}
