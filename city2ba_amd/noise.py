"""noise.rs free functions over the device-resident BAProblem (src/noise.rs:47-177, 388-416).

The reference consumes a BAProblem and returns a new one; here the problem is mutated on the device
and returned, which is the same thing under move semantics.  The reference draws from an unseeded
thread_rng(); these take an explicit `seed` (Philox4x32-10 counter RNG)."""
import ctypes as C

import numpy as np

from . import _lib as L


def _v3(v):
    a = np.ascontiguousarray(v, dtype=np.float64).reshape(3)
    return a, a.ctypes.data_as(C.c_void_p)


def add_drift(bal, strength, angle_strength, std, dir, seed=0):           # noqa: A002
    d, p = _v3(dir)
    L.check(L.lib().c2b_problem_add_drift(bal._h, float(strength), float(angle_strength), float(std), p, int(seed)))
    return bal


def add_drift_normalized(bal, strength, angle_strength, std, seed=0):
    L.check(L.lib().c2b_problem_add_drift_normalized(bal._h, float(strength), float(angle_strength), float(std),
                                                     int(seed)))
    return bal


def add_noise(bal, translation_std, rotation_std, point_std, observations_std, seed=0):
    L.check(L.lib().c2b_problem_add_noise(bal._h, float(translation_std), float(rotation_std), float(point_std),
                                          float(observations_std), int(seed)))
    return bal


def add_noise_with_errors(bal, translation_std, rotation_std, point_std, observations_std, seed=0):
    """add_noise followed by the L1 / L2 reprojection errors of the result -- run_noise's tail
    (src/bin/city2ba.rs:334-354) -- with the observation pass and both error sums in one launch.
    Returns (bal, l1, l2); the resident state is what add_noise alone leaves, bit for bit."""
    l1, l2 = C.c_double(), C.c_double()
    L.check(L.lib().c2b_problem_add_noise_errors_l1_l2(bal._h, float(translation_std), float(rotation_std), float(point_std),
                                                       float(observations_std), int(seed), C.byref(l1), C.byref(l2)))
    return bal, l1.value, l2.value


# ---- index-corruption functions (src/noise.rs:179-378): host-side reshuffles of the visibility graph ------------
def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _graph(bal):
    return (bal.cameras(), bal.points(), bal.row_ptr.astype(np.uint64).copy(), bal.pt_idx.astype(np.uint64).copy(),
            np.ascontiguousarray(bal.observations(), dtype=np.float64).copy())


def _rebuild(bal, cams, pts, row_ptr, pt_idx, uv):
    from .baproblem import BAProblem
    return BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv, bal._device)


def add_incorrect_correspondences(bal, mismatch_chance, seed=0):
    """add_incorrect_correspondences (src/noise.rs:180-226); returns a new problem"""
    cams, pts, row_ptr, pt_idx, uv = _graph(bal)
    L.check(L.lib().c2b_add_incorrect_correspondences(len(cams), _ptr(row_ptr), _ptr(pt_idx), _ptr(uv),
                                                      float(mismatch_chance), int(seed)))
    return _rebuild(bal, cams, pts, row_ptr, pt_idx, uv)


def drop_features(bal, drop_percent, seed=0):
    """drop_features (src/noise.rs:229-251): `drop_percent` is the fraction of each camera's observations KEPT
    (`l = len * drop_percent`, :238), as in the reference"""
    cams, pts, row_ptr, pt_idx, uv = _graph(bal)
    L.check(L.lib().c2b_drop_features(len(cams), _ptr(row_ptr), _ptr(pt_idx), _ptr(uv), float(drop_percent), int(seed)))
    n = int(row_ptr[-1])
    return _rebuild(bal, cams, pts, row_ptr, pt_idx[:n].copy(), uv[:n].copy())


def split_landmarks(bal, split_percent, seed=0):
    """split_landmarks (src/noise.rs:255-291)"""
    cams, pts, row_ptr, pt_idx, uv = _graph(bal)
    n_pts = len(pts)
    buf = np.zeros((n_pts + int(max(0.0, split_percent) * n_pts) + 1, 3))
    buf[:n_pts] = pts
    n = C.c_int64(n_pts)
    L.check(L.lib().c2b_split_landmarks(C.byref(n), _ptr(buf), len(buf), len(pt_idx), _ptr(pt_idx), float(split_percent),
                                        int(seed)))
    return _rebuild(bal, cams, buf[:n.value].copy(), row_ptr, pt_idx, uv)


def join_landmarks(bal, join_percent, seed=0):
    """join_landmarks (src/noise.rs:326-378)"""
    cams, pts, row_ptr, pt_idx, uv = _graph(bal)
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    L.check(L.lib().c2b_join_landmarks(len(pts), _ptr(pts), len(pt_idx), _ptr(pt_idx), float(join_percent), int(seed)))
    return _rebuild(bal, cams, pts, row_ptr, pt_idx, uv)


def add_sin_noise(ba, dir, noise_dir, strength, frequency):               # noqa: A002
    d, dp = _v3(dir)
    n, np_ = _v3(noise_dir)
    L.check(L.lib().c2b_problem_add_sin_noise(ba._h, dp, np_, float(strength), float(frequency)))
    return ba
