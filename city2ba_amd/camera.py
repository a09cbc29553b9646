"""SnavelyCamera's inherent methods (src/baproblem.rs:178-225) on batched camera records.

A camera is a row of 15 doubles, the reference's in-memory SnavelyCamera: `dir` (Basis3, column-major 3x3), `loc`,
`intrin` = (f, k1, k2).  from_vec / to_vec (Rodrigues both ways) run on the device through a temporary BAProblem;
the accessors are views."""
import numpy as np

from . import _lib as L


def _rows(cam15):
    a = np.asarray(cam15, dtype=np.float64)
    if a.shape[-1] != 15:
        raise L.City2baError(L.ERR_INVALID_ARGUMENT, "a camera record has 15 doubles (R col-major, t, f, k1, k2)")
    return a


def from_vec(bal9, device=0):
    """SnavelyCamera::from_vec (:180-186): 9-vectors (w, t, f, k1, k2) -> camera records [n,15]"""
    from .baproblem import BAProblem
    b = np.ascontiguousarray(bal9, dtype=np.float64).reshape(-1, 9)
    ba = BAProblem.from_bal(b, np.zeros((0, 3)), np.zeros(len(b) + 1, dtype=np.uint64), [], np.zeros((0, 2)), device)
    return ba.cameras()


def to_vec(cam15, device=0):
    """SnavelyCamera::to_vec (:189-202): camera records -> 9-vectors (w = to_rodrigues(dir))"""
    from .baproblem import BAProblem
    c = np.ascontiguousarray(_rows(cam15)).reshape(-1, 15)
    ba = BAProblem.from_visibility(c, np.zeros((0, 3)), np.zeros(len(c) + 1, dtype=np.uint64), [], np.zeros((0, 2)), device)
    return ba.cameras_bal()


def rotation(cam15):
    """rotation() (:205-207): the 3x3 matrices R (x_cam = R x + t)"""
    a = _rows(cam15)
    return np.swapaxes(a[..., :9].reshape(a.shape[:-1] + (3, 3)), -1, -2)      # stored column-major


def focal_length(cam15):
    """focal_length() (:209-211)"""
    return _rows(cam15)[..., 12]


def distortion(cam15):
    """distortion() (:213-215): (k1, k2)"""
    a = _rows(cam15)
    return a[..., 13], a[..., 14]


def modify_intrin(cam15, delta):
    """modify_intrin(delta) (:218-224): intrin + delta, everything else unchanged; returns a copy"""
    out = np.array(_rows(cam15), dtype=np.float64, copy=True)
    out[..., 12:15] = out[..., 12:15] + np.asarray(delta, dtype=np.float64).reshape(3)
    return out
