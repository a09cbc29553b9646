"""ctypes loader for the CPU oracle (oracle/city2ba_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (city2ba_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcity2ba_oracle.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64 = C.c_int64
_u64 = C.c_uint64
_d = C.c_double


def _digest():
    import hashlib
    h = hashlib.sha256()
    for name in ("city2ba_oracle.c", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False):
    """rebuilt when the SOURCE TEXT changed (a digest next to the library), not by time stamps: a snapshot of the tree on
    another machine does not keep their order"""
    stamp = _SO + ".stamp"
    try:
        with open(stamp) as fh:
            fresh = os.path.exists(_SO) and fh.read().strip() == _digest()
    except OSError:
        fresh = False
    if force or not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
        with open(stamp, "w") as fh:
            fh.write(_digest() + "\n")
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_from_rodrigues", None, _dp, _dp)
    sig("orc_to_rodrigues", None, _dp, _dp)
    sig("orc_camera_from_bal", None, _dp, _dp)
    sig("orc_camera_to_bal", None, _dp, _dp)
    sig("orc_project_world", None, _dp, _dp, _dp)
    sig("orc_project", None, _dp, _dp, _dp)
    sig("orc_pow4_cr", _d, _d)
    sig("orc_pow4_both", None, _dp, _i64, _dp, _dp)
    sig("orc_center", None, _dp, _dp)
    sig("orc_to_world", None, _dp, _dp, _dp)
    sig("orc_from_position_direction", None, _dp, _dp, _dp)
    sig("orc_transform", None, _dp, _dp, _dp, _dp)
    sig("orc_basis_from_angle_y_deg", None, _d, _dp)
    sig("orc_basis_from_angle_x_rad", None, _d, _dp)
    sig("orc_basis_from_axis_angle", None, _dp, _d, _dp)
    sig("orc_project_observations", None, _dp, _i64, _dp, _u64p, _u64p, _dp)
    sig("orc_total_reprojection_error", _d, _dp, _i64, _dp, _u64p, _u64p, _dp, _d)
    sig("orc_reprojection_error_sum", _d, _dp, _i64, _dp, _u64p, _u64p, _dp, _d)
    sig("orc_mean", None, _dp, _i64, _dp, _i64, _dp)
    sig("orc_std", None, _dp, _i64, _dp, _i64, _dp)
    sig("orc_extent", None, _dp, _i64, _dp, _i64, _dp, _dp)
    sig("orc_dimensions", None, _dp, _i64, _dp, _i64, _dp)
    sig("orc_visibility_pairs", None, _dp, _dp, _u32p, _u32p, _i64, _d, _dp, _u8p)
    sig("orc_occlusion_filter", None, _dp, _dp, _u32p, _u32p, _i64, _f32p, _i64, _u8p)
    sig("orc_residual_jacobian_one", None, _dp, _dp, _dp, _dp, _dp, _dp, _dp)
    sig("orc_residual_jacobian_bal", None, _dp, _i64, _dp, _u64p, _u64p, _dp, _dp, _dp, _dp)
    sig("orc_residual_jacobian", None, _dp, _i64, _dp, _u64p, _u64p, _dp, _dp, _dp, _dp)
    sig("orc_philox4x32_10", None, _u32p, _u32p, _u32p)
    sig("orc_philox2x32_10", None, _u32p, C.c_uint32, _u32p)
    sig("orc_normal_pair", None, _u64, C.c_uint32, _u64, C.c_uint32, _dp)
    sig("orc_drift_origin", None, _dp, _i64, _dp, _i64, _dp, C.POINTER(_i64))
    sig("orc_add_drift", None, _dp, _i64, _dp, _i64, _d, _d, _d, _dp, _u64)
    sig("orc_add_drift_normalized", None, _dp, _i64, _dp, _i64, _d, _d, _d, _u64)
    sig("orc_add_noise", None, _dp, _i64, _dp, _i64, _dp, _i64, _u64, _d, _d, _d, _d, _u64)
    sig("orc_add_sin_noise", None, _dp, _i64, _dp, _i64, _dp, _dp, _d, _d)
    sig("orc_bench_residual_jacobian", _d, _dp, _i64, _dp, _u64p, _u64p, _dp, _dp, _dp, _dp)
    sig("orc_visgraph_from_csr", C.c_void_p, _i64, _u64p, _u64p, _dp)
    sig("orc_visgraph_free", None, C.c_void_p)
    sig("orc_bench_run", _i64, C.c_int, C.c_int, _d, _i64, _dp, _i64, _dp, _u64p, _u64p, _dp, C.c_void_p, _d, _dp, _dp, _dp,
        C.POINTER(_d), C.POINTER(_d))
    _lib = L
    return L


def pow4_cr(x):
    return lib().orc_pow4_cr(float(x))


def pow4_both(x):
    """(libm pow(x, 4.0), correctly rounded x^4) for an array of x."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    a, b = np.empty_like(x), np.empty_like(x)
    lib().orc_pow4_both(x, x.size, a, b)
    return a, b


def _f(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


# ---- scalar-level helpers -------------------------------------------------
def from_rodrigues(w):
    R = np.empty(9)
    lib().orc_from_rodrigues(_f(w, 3), R)
    return R


def to_rodrigues(R):
    w = np.empty(3)
    lib().orc_to_rodrigues(_f(R, 9), w)
    return w


def camera_from_bal(bal9):
    bal9 = _f(bal9).reshape(-1, 9)
    out = np.empty((bal9.shape[0], 15))
    L = lib()
    for i in range(bal9.shape[0]):
        L.orc_camera_from_bal(bal9[i], out[i])
    return out


def camera_to_bal(cam15):
    cam15 = _f(cam15).reshape(-1, 15)
    out = np.empty((cam15.shape[0], 9))
    L = lib()
    for i in range(cam15.shape[0]):
        L.orc_camera_to_bal(cam15[i], out[i])
    return out


def project_world(cam15, p):
    q = np.empty(3)
    lib().orc_project_world(_f(cam15, 15), _f(p, 3), q)
    return q


def project(cam15, q):
    uv = np.empty(2)
    lib().orc_project(_f(cam15, 15), _f(q, 3), uv)
    return uv


def center(cam15):
    c = np.empty(3)
    lib().orc_center(_f(cam15, 15), c)
    return c


def centers(cams15):
    cams15 = _f(cams15).reshape(-1, 15)
    return np.stack([center(c) for c in cams15]) if len(cams15) else np.empty((0, 3))


def to_world(cam15, p):
    o = np.empty(3)
    lib().orc_to_world(_f(cam15, 15), _f(p, 3), o)
    return o


def from_position_direction(pos, R):
    cam = np.empty(15)
    lib().orc_from_position_direction(_f(pos, 3), _f(R, 9), cam)
    return cam


def transform(cam15, dR, dloc):
    out = np.empty(15)
    lib().orc_transform(_f(cam15, 15), _f(dR, 9), _f(dloc, 3), out)
    return out


def basis_from_angle_y_deg(deg):
    R = np.empty(9)
    lib().orc_basis_from_angle_y_deg(float(deg), R)
    return R


def basis_from_angle_x_rad(rad):
    R = np.empty(9)
    lib().orc_basis_from_angle_x_rad(float(rad), R)
    return R


def basis_from_axis_angle(ax, angle):
    R = np.empty(9)
    lib().orc_basis_from_axis_angle(_f(ax, 3), float(angle), R)
    return R


# ---- batch (CSR) ------------------------------------------------------------
def _csr(row_ptr, pt_idx):
    return (np.ascontiguousarray(row_ptr, dtype=np.uint64),
            np.ascontiguousarray(pt_idx, dtype=np.uint64))


def project_observations(cams15, pts, row_ptr, pt_idx):
    cams15 = _f(cams15).reshape(-1, 15)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    uv = np.empty((len(pt_idx), 2))
    lib().orc_project_observations(cams15, len(cams15), _f(pts), row_ptr, pt_idx, uv)
    return uv


def total_reprojection_error(cams15, pts, row_ptr, pt_idx, uv, norm):
    cams15 = _f(cams15).reshape(-1, 15)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    return lib().orc_total_reprojection_error(cams15, len(cams15), _f(pts), row_ptr, pt_idx,
                                              _f(uv), float(norm))


def reprojection_error_sum(cams15, pts, row_ptr, pt_idx, uv, norm):
    cams15 = _f(cams15).reshape(-1, 15)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    return lib().orc_reprojection_error_sum(cams15, len(cams15), _f(pts), row_ptr, pt_idx,
                                            _f(uv), float(norm))


def residual_jacobian(cams15, pts, row_ptr, pt_idx, uv):
    cams15 = _f(cams15).reshape(-1, 15)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    n = len(pt_idx)
    r, Jc, Jp = np.empty((n, 2)), np.empty((n, 18)), np.empty((n, 6))
    lib().orc_residual_jacobian(cams15, len(cams15), _f(pts), row_ptr, pt_idx, _f(uv), r, Jc, Jp)
    return r, Jc, Jp


def residual_jacobian_bal(bal9, pts, row_ptr, pt_idx, uv):
    """bal mode: columns refer to the 9-vector's own w; R = from_rodrigues(w)."""
    bal9 = _f(bal9).reshape(-1, 9)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    n = len(pt_idx)
    r, Jc, Jp = np.empty((n, 2)), np.empty((n, 18)), np.empty((n, 6))
    lib().orc_residual_jacobian_bal(bal9, len(bal9), _f(pts), row_ptr, pt_idx, _f(uv), r, Jc, Jp)
    return r, Jc, Jp


def residual_jacobian_one(cam15, w, X, uv_obs):
    r, Jc, Jp = np.empty(2), np.empty(18), np.empty(6)
    lib().orc_residual_jacobian_one(_f(cam15, 15), _f(w, 3), _f(X, 3), _f(uv_obs, 2), r, Jc, Jp)
    return r, Jc, Jp


def bench_residual_jacobian(cams15, pts, row_ptr, pt_idx, uv, r, Jc, Jp):
    cams15 = _f(cams15).reshape(-1, 15)
    row_ptr, pt_idx = _csr(row_ptr, pt_idx)
    return lib().orc_bench_residual_jacobian(cams15, len(cams15), _f(pts), row_ptr, pt_idx,
                                             _f(uv), r, Jc, Jp)


def bench_run(layout, threads, seconds, cams15, pts, row_ptr, pt_idx, uv, r, Jc, Jp, norm=2.0, max_passes=1000):
    """bench.py's cpu_baseline leg: whole passes of residual + Jacobian + error sum over the given cameras, timed inside
    C.  layout "faithful" = the reference's Vec<Vec<(usize,(f64,f64))>> storage and powf per term, "optimised" = flat
    CSR; threads = pthreads over equal contiguous camera ranges.  Returns (passes, seconds, error_sum)."""
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64); pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint64)
    uv = _f(uv)
    L = lib()
    graph = None
    if layout == "faithful":
        graph = L.orc_visgraph_from_csr(len(cams15), row_ptr, pt_idx, uv)
        assert graph, "out of memory building the Vec<Vec<>> graph"
    el, tot = _d(0.0), _d(0.0)
    try:
        passes = L.orc_bench_run(0 if layout == "faithful" else 1, int(threads), float(seconds), int(max_passes), cams15,
                                 len(cams15), pts, row_ptr, pt_idx, uv, graph, float(norm), r, Jc, Jp, C.byref(el), C.byref(tot))
    finally:
        if graph:
            L.orc_visgraph_free(graph)
    assert passes > 0
    return int(passes), el.value, tot.value


def mean(cams15, pts):
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    o = np.empty(3)
    lib().orc_mean(cams15, len(cams15), pts, len(pts), o)
    return o


def std(cams15, pts):
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    o = np.empty(3)
    lib().orc_std(cams15, len(cams15), pts, len(pts), o)
    return o


def extent(cams15, pts):
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    mn, mx = np.empty(3), np.empty(3)
    lib().orc_extent(cams15, len(cams15), pts, len(pts), mn, mx)
    return mn, mx


def dimensions(cams15, pts):
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    o = np.empty(3)
    lib().orc_dimensions(cams15, len(cams15), pts, len(pts), o)
    return o


def visibility_pairs(cams15, pts, cam_idx, pt_idx, max_dist):
    cams15 = _f(cams15).reshape(-1, 15)
    cam_idx = np.ascontiguousarray(cam_idx, dtype=np.uint32)
    pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    n = len(cam_idx)
    uv = np.empty((n, 2)); keep = np.empty(n, dtype=np.uint8)
    lib().orc_visibility_pairs(cams15, _f(pts), cam_idx, pt_idx, n, float(max_dist), uv, keep)
    return uv, keep


def occlusion_filter(cams15, pts, cam_idx, pt_idx, tri9):
    """keep mask of the occlusion rays of generate::visibility_graph (src/generate.rs:455-476), all triangles"""
    cams15 = _f(cams15).reshape(-1, 15)
    cam_idx = np.ascontiguousarray(cam_idx, dtype=np.uint32)
    pt_idx = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    tri = np.ascontiguousarray(tri9, dtype=np.float32).reshape(-1, 9)
    keep = np.empty(len(cam_idx), dtype=np.uint8)
    lib().orc_occlusion_filter(cams15, _f(pts), cam_idx, pt_idx, len(cam_idx), tri, len(tri), keep)
    return keep


# ---- noise --------------------------------------------------------------------
def philox4x32_10(ctr, key):
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(np.ascontiguousarray(ctr, dtype=np.uint32),
                            np.ascontiguousarray(key, dtype=np.uint32), out)
    return out


def philox2x32_10(ctr, key):
    out = np.empty(2, dtype=np.uint32)
    lib().orc_philox2x32_10(np.ascontiguousarray(ctr, dtype=np.uint32), int(key), out)
    return out


def normal_pair(seed, stream, entity, slot):
    z = np.empty(2)
    lib().orc_normal_pair(int(seed), int(stream), int(entity), int(slot), z)
    return z


def drift_origin(cams15, pts):
    cams15 = _f(cams15).reshape(-1, 15); pts = _f(pts).reshape(-1, 3)
    o = np.empty(3); idx = _i64(0)
    lib().orc_drift_origin(cams15, len(cams15), pts, len(pts), o, C.byref(idx))
    return o, idx.value


def add_drift(cams15, pts, strength, angle_strength, std_, dir_, seed):
    cams15 = _f(cams15).reshape(-1, 15).copy(); pts = _f(pts).reshape(-1, 3).copy()
    lib().orc_add_drift(cams15, len(cams15), pts, len(pts), strength, angle_strength, std_,
                        _f(dir_, 3), int(seed))
    return cams15, pts


def add_drift_normalized(cams15, pts, strength, angle_strength, std_, seed):
    cams15 = _f(cams15).reshape(-1, 15).copy(); pts = _f(pts).reshape(-1, 3).copy()
    lib().orc_add_drift_normalized(cams15, len(cams15), pts, len(pts), strength, angle_strength,
                                   std_, int(seed))
    return cams15, pts


def add_noise(cams15, pts, uv, translation_std, rotation_std, point_std, observations_std, seed,
              obs_offset=0):
    cams15 = _f(cams15).reshape(-1, 15).copy(); pts = _f(pts).reshape(-1, 3).copy()
    uv = _f(uv).reshape(-1, 2).copy()
    lib().orc_add_noise(cams15, len(cams15), pts, len(pts), uv, len(uv), int(obs_offset),
                        translation_std, rotation_std, point_std, observations_std, int(seed))
    return cams15, pts, uv


def add_sin_noise(cams15, pts, dir_, noise_dir, strength, frequency):
    cams15 = _f(cams15).reshape(-1, 15).copy(); pts = _f(pts).reshape(-1, 3).copy()
    lib().orc_add_sin_noise(cams15, len(cams15), pts, len(pts), _f(dir_, 3), _f(noise_dir, 3),
                            strength, frequency)
    return cams15, pts
