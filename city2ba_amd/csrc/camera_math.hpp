// camera_math.hpp -- device-side Snavely/BAL camera algebra for gfx950.
//
// Compiled with -ffp-contract=off: everything the reference defines (projection, center,
// transform, Rodrigues maps) is evaluated in the reference's operation order with separate
// IEEE mul/add/div/sqrt so that it rounds like the Rust CPU path (Rust never contracts).
// FMAs appear only where written explicitly (the build-defined Jacobian, Newton steps).
//
// Reference items restated: src/baproblem.rs:78-102 (Rodrigues maps), :141-175 (Camera
// methods).  cgmath 0.17 semantics (column-major Matrix3, trace-method quaternion, cofactor
// inverse, v * (1/|v|) normalisation) are the published algorithms of that crate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pow4_libm.hpp"

namespace c2b {

#define C2B_DEV __device__ __forceinline__

// 32 doubles = 256 B = two 128-B lines per record: line 0 holds everything projection needs (R, t, intrinsics: the
// light kernels read exactly one line per camera when the table is 256-B aligned), line 1 the rest of J_l and the
// centre.  (28-double records straddled lines: 1.75 lines per camera for the first 128 bytes.)
constexpr int kCamBlk = 32;       // doubles per camblk record (C2B_CAMBLK_DOUBLES)
constexpr int kCamHot = 24;       // leading doubles the per-observation kernels stage in LDS
// camblk offsets (of the logical 32-double record)
constexpr int kR = 0, kT = 9, kIntr = 12, kJl = 15, kCenter = 24;

// r05 (end of the round): WHERE the record's two lines sit.  The table is blocked in groups of kCamGroup = 8 cameras: a group's
// eight light lines (record doubles 0..15: R, t, intrinsics, J_l[0]) are contiguous (1 KB), its eight heavy lines (doubles 16..31:
// J_l[1..8], centre, pad) follow (1 KB).  The passes that need projection only then touch contiguous kilobytes and skip as many,
// instead of one 128-byte line of every 256: interleaved records cost the 256-MB Infinity Cache their whole 169 MB (it keeps more
// than the touched line), so the light passes' 230 MB of inputs thrashed it or not by where the arrays happened to lie -- 78 us
// or 90-112 us for the same projection pass back to back, by device (profiles/r05ar).  Measured layouts: blocks of 8 -> 78-80 us on
// every device; pairs (256 B + 256 B) and blocks of 64 (8 KB + 8 KB) -> 105-118 everywhere (those strides switch address-hash
// bits and halve the cache slices in use); blocks of 512 -> like the interleaved records.  The table keeps its shape --
// C2B_CAMBLK_DOUBLES per camera -- but holds WHOLE groups: allocate it for n_cam rounded up to a multiple of 8 cameras
// (cam_table_doubles); a table is prepared for its own cameras, never sliced by camera.
constexpr int kCamGroup = 8;
// A group's 256 doubles: [0, 128) eight light lines (record doubles 0..15) | [128, 192) eight J_l tails (doubles 16..23, 64 bytes each)
// | [192, 224) eight centres (doubles 24..27: x y z 0, 32 bytes each) | [224, 256) the rest of the pad (doubles 28..31).  So the Jacobian
// kernel touches the first 1.5 KB of a group, the visibility predicate 1 KB + 256 bytes, projection 1 KB -- not 2 KB each.
__host__ __device__ inline int cam_in_group_at(int k, int j) {          // record double j of the group's k-th camera, from the group's start
    return j < 16 ? k * 16 + j : (j < 24 ? 128 + k * 8 + (j - 16) : (j < 28 ? 192 + k * 4 + (j - 24) : 224 + k * 4 + (j - 28)));
}
__host__ __device__ inline int64_t cam_group_at(int64_t c) { return (c >> 3) * (int64_t)(kCamGroup * kCamBlk); }
__host__ __device__ inline int64_t cam_light_at(int64_t c) { return cam_group_at(c) + (c & 7) * 16; }
__host__ __device__ inline int64_t cam_center_at(int64_t c) { return cam_group_at(c) + 192 + (c & 7) * 4; }     // 3 doubles, 32-byte aligned
// record double j / 16-byte chunk j2 of camera c, as an offset in doubles from the table's start
__host__ __device__ inline int64_t cam_at(int64_t c, int j) { return cam_group_at(c) + cam_in_group_at((int)(c & 7), j); }
__host__ __device__ inline int64_t cam_chunk_at(int64_t c, int j2) { return cam_at(c, 2 * j2); }
__host__ __device__ inline int64_t cam_table_doubles(int64_t n_cam) { return ((n_cam + kCamGroup - 1) / kCamGroup * kCamGroup) * (int64_t)kCamBlk; }
// a camera's record read in place (global memory): rec[j] = record double j.  project_obs takes it like a pointer.
struct CamRec {
    const double *group; int k;
    __device__ __forceinline__ CamRec(const double *camblk, int64_t c) : group(camblk + cam_group_at(c)), k((int)(c & 7)) {}
    __device__ __forceinline__ double operator[](int j) const { return group[cam_in_group_at(k, j)]; }
};

constexpr double kEps = 2.220446049250313e-16;   // f64::EPSILON
constexpr double kPi = 3.14159265358979323846;

// ---- small vector helpers (cgmath evaluation order) ----------------------------------
// Templated on the scalar so that the f32 extension (BASELINE config 5) shares the algebra; the f64
// instantiations are what every parity statement refers to.
C2B_DEV void sincos_t(double a, double *s, double *c) { sincos(a, s, c); }
C2B_DEV void sincos_t(float a, float *s, float *c) { sincosf(a, s, c); }

template <typename T>
C2B_DEV T dot3(T ax, T ay, T az, T bx, T by, T bz) {
    return (ax * bx + ay * by) + az * bz;
}

// col-major 3x3 (cam15 state): m[3*c + r]
template <typename T>
C2B_DEV void cm_mat_vec(const T *m, T x, T y, T z, T o[3]) {
    o[0] = dot3(m[0], m[3], m[6], x, y, z);
    o[1] = dot3(m[1], m[4], m[7], x, y, z);
    o[2] = dot3(m[2], m[5], m[8], x, y, z);
}

// out = a * b, all col-major; out[c][r] = a.row(r) . b.col(c)
template <typename T>
C2B_DEV void cm_mat_mul(const T *a, const T *b, T *o) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[3 * c + r] = dot3(a[r], a[3 + r], a[6 + r], b[3 * c], b[3 * c + 1], b[3 * c + 2]);
}

// -(R^-1 t) with cgmath's Matrix3::invert (cross products / determinant), src/baproblem.rs:161-163
template <typename T>
C2B_DEV void cm_center(const T *m, T tx, T ty, T tz, T c[3]) {
    const T m00 = m[0], m01 = m[1], m02 = m[2];   // column 0 (c, r)
    const T m10 = m[3], m11 = m[4], m12 = m[5];   // column 1
    const T m20 = m[6], m21 = m[7], m22 = m[8];   // column 2
    const T det = m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02) +
                  m20 * (m01 * m12 - m11 * m02);
    // rows of the inverse: (c1 x c2)/det, (c2 x c0)/det, (c0 x c1)/det
    const T a0 = (m11 * m22 - m12 * m21) / det, a1 = (m12 * m20 - m10 * m22) / det,
            a2 = (m10 * m21 - m11 * m20) / det;
    const T b0 = (m21 * m02 - m22 * m01) / det, b1 = (m22 * m00 - m20 * m02) / det,
            b2 = (m20 * m01 - m21 * m00) / det;
    const T c0 = (m01 * m12 - m02 * m11) / det, c1 = (m02 * m10 - m00 * m12) / det,
            c2 = (m00 * m11 - m01 * m10) / det;
    c[0] = -dot3(a0, a1, a2, tx, ty, tz);
    c[1] = -dot3(b0, b1, b2, tx, ty, tz);
    c[2] = -dot3(c0, c1, c2, tx, ty, tz);
}

// Matrix3::from_axis_angle, col-major out
template <typename T>
C2B_DEV void cm_from_axis_angle(T ax, T ay, T az, T angle, T *o) {
    T s, c;
    sincos_t(angle, &s, &c);
    const T k = T(1.0) - c;
    o[0] = k * ax * ax + c;       o[1] = k * ax * ay + s * az;  o[2] = k * ax * az - s * ay;
    o[3] = k * ax * ay - s * az;  o[4] = k * ay * ay + c;       o[5] = k * ay * az + s * ax;
    o[6] = k * ax * az + s * ay;  o[7] = k * ay * az - s * ax;  o[8] = k * az * az + c;
}

// From<Matrix3> for Quaternion (trace method); q = {s, x, y, z}; m col-major
C2B_DEV void cm_quat_from_mat(const double *m, double q[4]) {
    const double m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5],
                 m20 = m[6], m21 = m[7], m22 = m[8];
    const double trace = m00 + m11 + m22;
    double w, x, y, z, s;
    if (trace >= 0.0) {
        s = sqrt(1.0 + trace);
        w = 0.5 * s; s = 0.5 / s;
        x = (m12 - m21) * s; y = (m20 - m02) * s; z = (m01 - m10) * s;
    } else if (m00 > m11 && m00 > m22) {
        s = sqrt((m00 - m11 - m22) + 1.0);
        x = 0.5 * s; s = 0.5 / s;
        y = (m10 + m01) * s; z = (m02 + m20) * s; w = (m12 - m21) * s;
    } else if (m11 > m22) {
        s = sqrt((m11 - m00 - m22) + 1.0);
        y = 0.5 * s; s = 0.5 / s;
        z = (m21 + m12) * s; x = (m10 + m01) * s; w = (m20 - m02) * s;
    } else {
        s = sqrt((m22 - m00 - m11) + 1.0);
        z = 0.5 * s; s = 0.5 / s;
        x = (m02 + m20) * s; y = (m21 + m12) * s; w = (m01 - m10) * s;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

// From<Quaternion> for Matrix3, col-major out
C2B_DEV void cm_mat_from_quat(const double q[4], double *o) {
    const double s = q[0], x = q[1], y = q[2], z = q[3];
    const double x2 = x + x, y2 = y + y, z2 = z + z;
    const double xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
    const double yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    const double sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    o[0] = 1.0 - yy2 - zz2; o[1] = xy2 + sz2;       o[2] = xz2 - sy2;
    o[3] = xy2 - sz2;       o[4] = 1.0 - xx2 - zz2; o[5] = yz2 + sx2;
    o[6] = xz2 + sy2;       o[7] = yz2 - sx2;       o[8] = 1.0 - xx2 - yy2;
}

// from_rodrigues, src/baproblem.rs:78-90
C2B_DEV void from_rodrigues(double w0, double w1, double w2, double *Rcm) {
    const double theta2 = dot3(w0, w1, w2, w0, w1, w2);
    if (theta2 > kEps) {
        const double angle = sqrt(theta2);
        const double inv = 1.0 / angle;
        cm_from_axis_angle(w0 * inv, w1 * inv, w2 * inv, angle, Rcm);
    } else {
        const double m[9] = {1.0, w2, -w1, -w2, 1.0, w0, w1, -w0, 1.0};
        double q[4];
        cm_quat_from_mat(m, q);
        cm_mat_from_quat(q, Rcm);
    }
}

// to_rodrigues, src/baproblem.rs:93-102
C2B_DEV void to_rodrigues(const double *Rcm, double w[3]) {
    double q[4];
    cm_quat_from_mat(Rcm, q);
    const double angle = 2.0 * acos(q[0]);
    const double one_m = 1.0 - q[0] * q[0];
    if (one_m < kEps) {
        w[0] = w[1] = w[2] = 0.0;
    } else {
        const double d = sqrt(one_m);
        const double ax = q[1] / d, ay = q[2] / d, az = q[3] / d;
        const double inv = 1.0 / sqrt(dot3(ax, ay, az, ax, ay, az));
        w[0] = (ax * inv) * angle; w[1] = (ay * inv) * angle; w[2] = (az * inv) * angle;
    }
}

// Camera::transform, src/baproblem.rs:165-171 (new loc uses the OLD dir), in place on cam15
template <typename T>
C2B_DEV void transform_cam15(T *cam, const T *dRcm, T dx, T dy, T dz) {
    T c[3], v[3], nr[9];
    cm_center(cam, cam[9], cam[10], cam[11], c);
    cm_mat_vec(cam, c[0] + dx, c[1] + dy, c[2] + dz, v);
    cm_mat_mul(cam, dRcm, nr);
#pragma unroll
    for (int i = 0; i < 9; ++i) cam[i] = nr[i];
    cam[9] = T(-1.0) * v[0]; cam[10] = T(-1.0) * v[1]; cam[11] = T(-1.0) * v[2];
}

// Left Jacobian of SO(3), J_l(w) = I + a [w]x + b [w]x^2 (row-major out).  Build-defined
// (feeds the Jacobian's rotation columns); series below |w| = 0.1 to avoid cancellation.
C2B_DEV void left_jacobian(double w0, double w1, double w2, double *J) {
    const double t2 = (w0 * w0 + w1 * w1) + w2 * w2;
    double a, b;
    if (t2 < 1e-2) {
        a = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800))));
        b = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800))));
    } else {
        const double t = sqrt(t2);
        double sh, ch, s, c;
        sincos(0.5 * t, &sh, &ch);
        sincos(t, &s, &c);
        (void)ch; (void)c;
        a = 2.0 * sh * sh / t2;
        b = (t - s) / (t2 * t);
    }
    // K = [w]x ; K^2 = w w^T - t2 I
    J[0] = 1.0 + b * (w0 * w0 - t2); J[1] = -a * w2 + b * w0 * w1;    J[2] = a * w1 + b * w0 * w2;
    J[3] = a * w2 + b * w0 * w1;     J[4] = 1.0 + b * (w1 * w1 - t2); J[5] = -a * w0 + b * w1 * w2;
    J[6] = -a * w1 + b * w0 * w2;    J[7] = a * w0 + b * w1 * w2;     J[8] = 1.0 + b * (w2 * w2 - t2);
}

// Fill one camblk record from the state (col-major R, t, intrin) and the Rodrigues vector the
// Jacobian columns refer to.
C2B_DEV void fill_camblk(const double *cam15, double w0, double w1, double w2, double *blk) {
    // row-major R
    blk[0] = cam15[0]; blk[1] = cam15[3]; blk[2] = cam15[6];
    blk[3] = cam15[1]; blk[4] = cam15[4]; blk[5] = cam15[7];
    blk[6] = cam15[2]; blk[7] = cam15[5]; blk[8] = cam15[8];
#pragma unroll
    for (int i = 0; i < 6; ++i) blk[9 + i] = cam15[9 + i];
    left_jacobian(w0, w1, w2, blk + kJl);
    double c[3];
    cm_center(cam15, cam15[9], cam15[10], cam15[11], c);
    blk[24] = c[0]; blk[25] = c[1]; blk[26] = c[2];
#pragma unroll
    for (int i = 27; i < kCamBlk; ++i) blk[i] = 0.0;
}

// ---- per-observation hot path ----------------------------------------------------------
struct Proj {
    double qx, qy, qz;     // camera-frame point  (project_world, src/baproblem.rs:141-143)
    double px, py;         // -q.xy / q.z
    double n, rad;         // |p|^2 and 1 + k1 n + k2 n^2
    double u, v;           // pixel             (project, src/baproblem.rs:145-151)
};

// |p|^4 the reference's way: libm's pow(|p|, 4.0), restated from glibc's machine code in pow4_libm.hpp (bit for bit the
// image's pow on every argument class; tests/test_pow4.py, tests/test_gpu_parity.py).  The two tables (4 KB + 2 KB) live
// in global memory: the rows a wave touches stay in the vector L1 / L2, and only cameras with k2 != 0 ever read them.
// (Rounds 1-5 evaluated the correctly rounded x^4 here; the CPU checker had a second mode to match.  Both are gone.)
__device__ const PowLogRow g_pow_log_tab[C2B_POW_LOG_ROWS] __attribute__((aligned(32))) = {C2B_POW_LOG_TAB};
__device__ const PowExpRow g_pow_exp_tab[C2B_POW_EXP_ROWS] __attribute__((aligned(16))) = {C2B_EXP_TAB};
C2B_DEV double pow4_libm(double x) { return pow4_glibc(x, g_pow_log_tab, g_pow_exp_tab); }

// q1 = a1 / b and q2 = a2 / b, both IEEE-correct and bit-identical to the compiler's own expansion of `/` (the
// v_div_scale / v_rcp / 4 FMA / v_div_fmas / v_div_fixup sequence, restated with the same builtins in the same order),
// with the refined reciprocal of the shared denominator computed once: 16 instructions and one v_rcp_f64 instead of 22
// and two.  v_div_scale's scaled denominator depends on the numerator only in the extreme-exponent cases it exists for;
// if the two numerators would scale the denominator differently the second quotient takes the ordinary path.
C2B_DEV void div2_shared(double a1, double a2, double b, double &q1, double &q2) {
    bool f1, f2, fd;
    const double d = __builtin_amdgcn_div_scale(a1, b, false, &fd);        // scaled denominator (w.r.t. a1)
    const double d2 = __builtin_amdgcn_div_scale(a2, b, false, &fd);       // ... w.r.t. a2
    double r = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, r, 1.0);
    r = fma(r, e0, r);
    const double e1 = fma(-d, r, 1.0);
    r = fma(r, e1, r);
    const double n1 = __builtin_amdgcn_div_scale(a1, b, true, &f1);        // scaled numerator + the fmas flag
    const double m1 = n1 * r;
    q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(fma(-d, m1, n1), r, m1, f1), b, a1);
    if (__builtin_expect(__double_as_longlong(d2) == __double_as_longlong(d), 1)) {
        const double n2 = __builtin_amdgcn_div_scale(a2, b, true, &f2);
        const double m2 = n2 * r;
        q2 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(fma(-d, m2, n2), r, m2, f2), b, a2);
    } else {
        q2 = a2 / b;
    }
}

// P is a pointer to double in any address space: a generic pointer, or one typed LDS-only / global-only (lds_cptr /
// glb_cptr) so that the loads compile to ds_read / global_load and two call sites can never be merged into FLAT loads.
typedef const __attribute__((address_space(3))) double *lds_cptr;
typedef const __attribute__((address_space(1))) double *glb_cptr;

// The projection in the pieces the kernels assemble it from.  `cam` points at a camblk-shaped record (LDS, global, or anything
// indexable by record double).  The reference writes |p|^4 as p.magnitude().powf(4.0) (src/baproblem.rs:147-149) = libm's
// pow(sqrt(n), 4.0): pow4_libm above.
//   project_head: project_world + the perspective divide + |p|^2;
//   project_tail: the radial factor and the pixel, |p|^4 handed in;
//   project_obs_k0: head + tail with |p|^4 = n * n -- FINAL when k2 == 0 (k2 * n^4 = 0 whatever the rounding of n^4), and what
//     every observation of the generators' default cameras takes: straight-line code, nothing of pow() in it;
//   project_obs: the reference's projection for any k2 -- |p|^4 = libm's pow(|p|, 4.0) (pow4_libm) when k2 != 0.
// The per-observation kernels run project_obs_k0 on everything and, behind ONE wave-uniform branch on "some lane's k2 != 0",
// project_obs again for those lanes (kernels.hpp): the pow4 code is register-hungry and they have no registers to spare, so it
// runs where next to nothing else is live, and the common path is exactly the code of rounds 1-5.
template <typename P>
C2B_DEV Proj project_head(P cam, double X, double Y, double Z) {
    Proj p;
    p.qx = dot3(cam[0], cam[1], cam[2], X, Y, Z) + cam[9];
    p.qy = dot3(cam[3], cam[4], cam[5], X, Y, Z) + cam[10];
    p.qz = dot3(cam[6], cam[7], cam[8], X, Y, Z) + cam[11];
    div2_shared(-p.qx, -p.qy, p.qz, p.px, p.py);            // -q.x / q.z, -q.y / q.z (src/baproblem.rs:146)
    p.n = p.px * p.px + p.py * p.py;
    return p;
}
C2B_DEV void project_tail(Proj &p, double f, double k1, double k2, double n4) {
    p.rad = 1.0 + k1 * p.n + k2 * n4;
    const double fr = f * p.rad;
    p.u = fr * p.px;
    p.v = fr * p.py;
}
template <typename P>
C2B_DEV Proj project_obs_k0(P cam, double X, double Y, double Z) {
    Proj p = project_head(cam, X, Y, Z);
    project_tail(p, cam[12], cam[13], cam[14], p.n * p.n);
    return p;
}
template <typename P>
C2B_DEV Proj project_obs(P cam, double X, double Y, double Z) {
    Proj p = project_head(cam, X, Y, Z);
    const double k2 = cam[14];
    double n4 = p.n * p.n;
    if (k2 != 0.0) n4 = pow4_libm(sqrt(p.n));
    project_tail(p, cam[12], cam[13], k2, n4);
    return p;
}

// |x|^norm with the two norms the reference's callers use special-cased (exact for both).
C2B_DEV double abs_pow(double x, double norm) {
    const double a = fabs(x);
    if (norm == 2.0) return a * a;
    if (norm == 1.0) return a;
    return pow(a, norm);
}
// The same with the norm fixed at compile time (the launchers dispatch on the host-side value): kernels for the
// reference's two norms carry no pow() code at all.  NORM_ANY = any other exponent.
enum { NORM_ANY = 0, NORM_1 = 1, NORM_2 = 2 };
C2B_DEV double pow_lean(double a, double y);      // below, with the other lean transcendentals
template <int NK>
C2B_DEV double abs_pow_k(double x, double norm) {
    const double a = fabs(x);
    if (NK == NORM_2) return a * a;
    if (NK == NORM_1) return a;
    return pow_lean(a, norm);
}

// ---- Philox4x32-10 + Box-Muller (build-defined draw scheme) -------------
enum : uint32_t { kStreamDriftCam = 1, kStreamDriftPt = 2, kStreamNoiseCam = 3,
                  kStreamNoisePt = 4, kStreamNoiseObs = 5 };

C2B_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- lean transcendentals for the noise draws ---------------------------------------------------------------------
// (the ENTITY draws -- normal_pair -- and pow_lean use these; the observation draw moved to tables in r05, further down)
// k_add_noise_observations is bound by vector issue (SQ counters, profiles/r03d: ACTIVE_INST_VALU 85 % of the SIMDs'
// cycles, 287 vector instructions per wave of 64 observations), and half of those instructions were the library's
// log / sincospi / cospi, which carry double-double arithmetic and argument handling this kernel has no use for: its
// log argument is a normal number in (0, 1], its angles are exact 32-bit fractions of a turn.  The forms below are the
// classic fdlibm kernels (Sun, 1993: e_log.c, k_sin.c, k_cos.c; errors below 1 ulp on their intervals) on exactly
// those domains, with explicit FMAs.  tests/test_gpu_parity.py pins them against libm to 1e-13 (the draws' tolerance
// is 1e-9).

// ln(x) for a normal x > 0 (the noise draws call it on (0, 1], pow_lean on any finite positive number)
C2B_DEV double log_unit(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);                   // x = m * 2^e, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 0.70710678118654752440) { m = m + m; e -= 1; }   // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);                      // 1 / d: hardware estimate + two Newton steps
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double q = f * r;
    q = fma(fma(-d, q, f), r, q);                            // q = f / (2 + f)
    const double z = q * q, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, de = (double)e;
    return fma(de, 6.93147180369123816490e-01, -((hfsq - fma(q, hfsq + R, de * 1.90821492927058770002e-10)) - f));
}

// exp(x), fdlibm e_exp.c: k = nearest integer to x / ln 2, r = x - k ln 2 in two pieces, a degree-5 polynomial in r^2,
// scaled by 2^k (v_ldexp_f64 saturates to 0 / inf by itself)
C2B_DEV double exp_lean(double x) {
    const double kf = rint(x * 1.44269504088896338700e+00);
    const double hi = fma(-kf, 6.93147180369123816490e-01, x), lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo, t = r * r;
    const double c = r - t * fma(t, fma(t, fma(t, fma(t, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                                6.61375632143793436117e-05), -2.77777777770155933842e-03), 1.66666666666666019037e-01);
    const double d = 2.0 - c;
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    q = fma(fma(-d, q, 1.0), q, q);
    const double y = 1.0 - ((lo - (r * c) * q) - hi);
    return ldexp(y, (int)kf);
}

// a^y for the error norms other than 1 and 2 (|residual|^norm, src/baproblem.rs:273-276): exp(y ln a) through the two
// kernels above for finite a > 0 -- relative error ~ |y ln a| ulps (1e-14 at |y ln a| = 45), against a sum compared at
// 1e-12 --, the library's pow for everything else (0, inf, NaN, subnormal results; a branch no sane residual takes).
// The library's pow made the NORM_ANY error kernel 2.7 times slower than its L1 / L2 instances (r02: 0.22 of the peak).
C2B_DEV double pow_lean(double a, double y) {
    const double l = y * log_unit(a);
    if (!(a > 0x1.0p-1000 && a < 0x1.0p+1000) || !(l > -600.0 && l < 600.0)) return pow(a, y);
    return exp_lean(l);
}

C2B_DEV double pow_t(double a, double y) { return pow_lean(a, y); }      // add_drift's distance^1.2 (src/noise.rs:104)
C2B_DEV float pow_t(float a, float y) { return powf(a, y); }

// sin and cos of x in [-pi/4, pi/4] (fdlibm __kernel_sin / __kernel_cos without the tail argument)
C2B_DEV void sincos_kernel(double x, double &sn, double &cs) {
    const double z = x * x;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                  2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    sn = fma(x * z, ps, x);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                  -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    cs = fma(z * z, pc, fma(-0.5, z, 1.0));
}

// (sin, cos) of 2 pi u for u in [0, 1) given as a double (a 53-bit fraction of a turn): 4u, its nearest integer and the
// remainder are all exact in double arithmetic, so the reduction to [-pi/4, pi/4] costs one rounding (the product with
// pi/2)
C2B_DEV void sincos_turns(double u, double &sn, double &cs) {
    const double t = 4.0 * u, kf = rint(t);
    const int k = (int)kf;                                            // 0 .. 4
    double s, c;
    sincos_kernel((t - kf) * 1.57079632679489661923, s, c);
    const bool swap = (k & 1) != 0;
    const double a = swap ? s : c, b = swap ? c : s;
    cs = ((k + 1) & 2) ? -a : a;
    sn = (k & 2) ? -b : b;
}

// two independent N(0, 1) for (seed; stream, entity, slot): Philox4x32-10 block -> u1 in (0, 1], u2 in [0, 1) ->
// Box-Muller.  Same draws as rounds 1-2; since r03 through the lean log / sincos above (the library's carried
// double-double arithmetic and argument handling these domains do not need; agreement with libm ~1e-15).
C2B_DEV void normal_pair(uint64_t seed, uint32_t stream, uint64_t entity, uint32_t slot,
                         double &z0, double &z1) {
    uint32_t o[4];
    philox4x32_10((uint32_t)entity, (uint32_t)(entity >> 32), slot, stream, (uint32_t)seed,
                  (uint32_t)(seed >> 32), o);
    const uint64_t a = ((uint64_t)o[1] << 32) | o[0];
    const uint64_t b = ((uint64_t)o[3] << 32) | o[2];
    const double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;   // (0,1]
    const double u2 = (double)(b >> 11) * 0x1.0p-53;         // [0,1)
    const double rad = sqrt(-2.0 * log_unit(u1));
    double s, c;
    sincos_turns(u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

// Philox2x32-10 (Salmon et al., SC'11; Random123's philox2x32_R(10, ...)): one 32 x 32 -> 64 multiply per round where
// Philox4x32 has two -- the multiplies run at a quarter of the vector rate, and with its xors Philox4x32-10 was 42 % of
// the observation-noise kernel's issue cycles (19 v_mad_u64_u32 + 40 v_xor of 214 vector instructions, r03 ISA).
// (gfx950 has no v_xor3_b32 -- the assembler rejects it --, so a round is one multiply and two xors.)
C2B_DEV void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t out[2]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p = (uint64_t)0xD256D193u * c0;
        const uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}

// ---- the observation draw's tables (r05) ---------------------------------------------------------------------------
// SQ counters (profiles/r05h_light_sq.json) put the two observation-noise passes at 69-75 % vector issue, 186 of their
// instructions being this draw; 80 of those were three sine / cosine polynomials and a logarithm with its division.  Both
// angles are 16-bit fractions of a turn and the logarithm's argument has 32 significant bits, so: (cos, sin)(2 pi a / 65536) =
// the angle-sum of two 256-entry tables (high byte, low byte: two fused multiply-adds each), and ln through 128 intervals
// of the mantissa (r = z / c - 1 by one FMA against the tabled 1 / c, a degree-7 series in |r| < 2^-7, the tabled ln c) --
// the form of ARM's optimised-routines log, with the two intervals touching z = 1 given c = 1 so that draws next to u = 1
// keep their relative accuracy.  640 16-byte entries (tools/gen_noise_tables.py: correctly rounded from 60 digits,
// tests/test_noise_tables.py), staged in LDS by the workgroup.  Against glibc: directions within 1.2e-16 absolute,
// -2 ln u within 3 ulps, its square root within 2.3e-16 relative (tests/test_gpu_parity.py pins the draw at 1e-13).
constexpr int kNoiseTab = 640;                 // double2 entries the observation draw stages: 10 240 bytes
constexpr int kNoiseTabLog = 512;
constexpr int kPowTab = 704;                   // ... and with the exponential's 128 doubles behind them (pow_tab): 11 264 bytes
typedef double tab2_t __attribute__((ext_vector_type(2)));      // (the class type double2 cannot be read through an LDS-typed pointer)
__device__ const tab2_t g_noise_tab[kPowTab] = {
#include "noise_tables.inc"
};
typedef const __attribute__((address_space(3))) tab2_t *lds_tab;

// every thread of the workgroup calls this, then a workgroup barrier
C2B_DEV void noise_tab_stage(tab2_t *sTab, int tid, int n_threads) {
    for (int i = tid; i < kNoiseTab; i += n_threads) sTab[i] = g_noise_tab[i];
}

// -2 ln(x) for a normal x in (0, 1]
C2B_DEV double m2log_tab(double x, lds_tab tab) {
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const uint32_t tmp = hi - 0x3FE60000u;                                // x = z 2^k, z in [0.6875, 1.375)
    const uint32_t i = (tmp >> 13) & 127u;
    const int k = (int32_t)tmp >> 20;
    const double z = __hiloint2double((int)(hi - (tmp & 0xFFF00000u)), __double2loint(x));
    const tab2_t e = tab[kNoiseTabLog + i];                               // (1 / c, -2 ln c)
    const double r = fma(z, e.x, -1.0), kd = (double)k;
    // -2 ln(1 + r) = -2 r + r^2 (1 + r (-2/3 + r (1/2 + r (-2/5 + r (1/3 - 2/7 r)))))
    double p = fma(r, -2.0 / 7.0, 1.0 / 3.0);
    p = fma(r, p, -2.0 / 5.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, -2.0 / 3.0);
    p = fma(r, p, 1.0);
    const double head = fma(kd, -2.0 * 6.93147180369123816490e-01, e.y);   // k ln2_hi is exact (ln2_hi ends in 21 zero bits)
    const double tail = fma(kd, -2.0 * 1.90821492927058770002e-10, -2.0 * r);
    return head + fma(r * r, p, tail);
}

// sqrt(x) for x = 0 or a normal x >= 2^-40 (what m2log_tab returns): hardware estimate of 1 / sqrt(x), one coupled
// Goldschmidt step, two residual corrections -- without the scaling and class tests the library's sqrt carries for
// subnormal and infinite arguments
C2B_DEV double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// exp(x) for |x| < 700: x = (128 e + j) ln2 / 128 + r, |r| <= ln2 / 256; 2^(j / 128) from the table, a degree-5 series in r,
// the exponent by v_ldexp_f64 (ln2 / 128 in two pieces, the high one short enough for k ln2_hi to be exact)
C2B_DEV double exp_tab(double x, lds_tab tab) {
    const double kf = rint(x * 0x1.71547652b82fep+7);
    double r = fma(-kf, 0x1.62e42fee00000p-8, x);
    r = fma(-kf, 0x1.a39ef35793c76p-40, r);
    const int ki = (int)kf;
    const double T = reinterpret_cast<const __attribute__((address_space(3))) double *>(tab + kNoiseTab)[ki & 127];
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(r, q, 1.0 / 6.0);
    q = fma(r, q, 0.5);
    const double p = fma(r * r, q, r);
    return ldexp(fma(T, p, T), ki >> 7);
}

// a^y through the two tables for finite a in (2^-1000, 2^1000) with |y ln a| < 600, the library's pow for everything else --
// pow_lean's contract (relative error ~ |y ln a| ulps) at 38 instead of 70 vector instructions.  `tab` holds kPowTab entries.
C2B_DEV double pow_tab(double a, double y, lds_tab tab) {
    const double l = (-0.5 * y) * m2log_tab(a, tab);
    if (!(a > 0x1.0p-1000 && a < 0x1.0p+1000) || !(l > -600.0 && l < 600.0)) return pow(a, y);
    return exp_tab(l, tab);
}
// |x|^norm for a norm other than 1 and 2 (abs_pow_k<NORM_ANY> with the tables at hand)
C2B_DEV double abs_pow_tab(double x, double norm, lds_tab tab) { return pow_tab(fabs(x), norm, tab); }

// (cos, sin) of a / 65536 turns, a < 65536
C2B_DEV void sincos_turns16(uint32_t a, lds_tab tab, double &cs, double &sn) {
    const tab2_t A = tab[a >> 8], B = tab[256 + (a & 255u)];
    cs = fma(A.x, B.x, -(A.y * B.y));
    sn = fma(A.y, B.x, A.x * B.y);
}

// add_noise's observation draw (src/noise.rs:152-170: a uniformly distributed unit 2-vector times Normal(0, std)) from
// ONE Philox2x32-10 block (r04; rounds 1-2 spent two Philox4x32 blocks here, r03 one): counter = the observation's global
// index, its high word xor-ed with the seed's high word; key = the seed's low word.  Word 0 -> the radius uniform
// u1 = (w0 + 1) 2^-32 in (0, 1] (the magnitude's tail ends at sqrt(-2 ln 2^-32) = 6.7 sigma); word 1, high half -> the
// Box-Muller angle, low half -> the direction (16-bit fractions of a turn each).  The normalised Gaussian pair of
// unit_random has a uniform direction and its radius cancels, so only the direction is drawn.  The CPU restatement
// under oracle/ reads the same bits (and evaluates them with libm).  c, s = the direction; returns the standard normal z.
// `tab` = the workgroup's LDS copy of g_noise_tab.
C2B_DEV double obs_noise_draw(uint64_t seed, uint64_t observation, lds_tab tab, double &c, double &s) {
    uint32_t o[2];
    philox2x32_10((uint32_t)observation, (uint32_t)(observation >> 32) ^ (uint32_t)(seed >> 32), (uint32_t)seed, o);
    const double u1 = fma((double)o[0], 0x1.0p-32, 0x1.0p-32);          // exact: (w0 + 1) 2^-32
    sincos_turns16(o[1] & 0xffffu, tab, c, s);
    double ca, sa;
    sincos_turns16(o[1] >> 16, tab, ca, sa);
    return sqrt_pos(m2log_tab(u1, tab)) * ca;
}

}  // namespace c2b
