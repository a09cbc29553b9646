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


def add_sin_noise(ba, dir, noise_dir, strength, frequency):               # noqa: A002
    d, dp = _v3(dir)
    n, np_ = _v3(noise_dir)
    L.check(L.lib().c2b_problem_add_sin_noise(ba._h, dp, np_, float(strength), float(frequency)))
    return ba
