#!/usr/bin/env python3
"""Generate tests/golden/snavely_golden.json with mpmath (60 significant digits).

The reference (Rust) cannot be executed in the build image, and it holds no numeric
golden vectors for this path.  These vectors therefore come from an
independent high-precision evaluation of the *mathematical* BAL/Snavely model that
src/baproblem.rs:141-151 and the doc block :553-579 implement:

    q  = R(w) X + t,        R(w) = exp([w]x)  (Rodrigues)
    p  = -q.xy / q.z
    uv = f (1 + k1 |p|^2 + k2 |p|^4) p

Jacobians d(uv)/d(w,t,f,k1,k2) (2x9) and d(uv)/dX (2x3) are obtained by mpmath's
high-order numerical differentiation (mp.diff) of that model -- i.e. without using any
hand-derived formula -- and cross-checked here against an analytic derivation.

Inputs are exact float64 values (stored with repr round-trip); outputs are the
correctly-rounded float64 of the 60-digit result.

Run:  python tests/golden/gen_golden.py      (takes about a minute)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))


def skew(w):
    return mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def rot(w):
    w = [mp.mpf(x) for x in w]
    th2 = w[0] ** 2 + w[1] ** 2 + w[2] ** 2
    K = skew(w)
    if th2 == 0:
        return mp.eye(3)
    th = mp.sqrt(th2)
    return mp.eye(3) + (mp.sin(th) / th) * K + ((1 - mp.cos(th)) / th2) * (K * K)


def model(params, X):
    """params = 9 BAL values, X = 3; returns (q, uv) as mp."""
    w, t, f, k1, k2 = params[0:3], params[3:6], params[6], params[7], params[8]
    R = rot(w)
    Xv = mp.matrix([mp.mpf(x) for x in X])
    q = R * Xv + mp.matrix([mp.mpf(x) for x in t])
    px, py = -q[0] / q[2], -q[1] / q[2]
    n = px * px + py * py
    r = 1 + mp.mpf(k1) * n + mp.mpf(k2) * n * n
    return q, (mp.mpf(f) * r * px, mp.mpf(f) * r * py)


def jac_numeric(params, X):
    """2x12 Jacobian by mp.diff (no hand derivation)."""
    allp = [mp.mpf(float(x)) for x in list(params) + list(X)]
    J = [[None] * 12 for _ in range(2)]
    for j in range(12):
        for i in range(2):
            def fn(v, j=j, i=i):
                a = list(allp)
                a[j] = v
                return model(a[:9], a[9:])[1][i]
            J[i][j] = mp.diff(fn, allp[j])
    return J


def jac_analytic(params, X):
    """Independent analytic form (left-Jacobian of SO(3)); used only as a cross-check."""
    w = [mp.mpf(float(x)) for x in params[0:3]]
    f, k1, k2 = (mp.mpf(float(x)) for x in params[6:9])
    R = rot(w)
    q, _ = model([mp.mpf(float(x)) for x in params], [mp.mpf(float(x)) for x in X])
    t = mp.matrix([mp.mpf(float(x)) for x in params[3:6]])
    y = q - t
    th2 = w[0] ** 2 + w[1] ** 2 + w[2] ** 2
    K = skew(w)
    if th2 == 0:
        Jl = mp.eye(3)
    else:
        th = mp.sqrt(th2)
        Jl = mp.eye(3) + ((1 - mp.cos(th)) / th2) * K + ((th - mp.sin(th)) / (th2 * th)) * (K * K)
    D = -skew(y) * Jl
    px, py = -q[0] / q[2], -q[1] / q[2]
    n = px * px + py * py
    r = 1 + k1 * n + k2 * n * n
    c = 2 * k1 + 4 * k2 * n
    B = mp.matrix([[f * (r + c * px * px), f * c * px * py], [f * c * px * py, f * (r + c * py * py)]])
    P = mp.matrix([[-1 / q[2], 0, q[0] / q[2] ** 2], [0, -1 / q[2], q[1] / q[2] ** 2]])
    A = B * P
    Jw = A * D
    JX = A * R
    J = [[None] * 12 for _ in range(2)]
    p = [px, py]
    for i in range(2):
        for j in range(3):
            J[i][j] = Jw[i, j]
            J[i][3 + j] = A[i, j]
            J[i][9 + j] = JX[i, j]
        J[i][6] = r * p[i]
        J[i][7] = f * n * p[i]
        J[i][8] = f * n * n * p[i]
    return J


def make_pair(rng, kind):
    if kind == "zero":
        w = np.zeros(3)
    elif kind == "tiny9":
        w = rng.normal(size=3); w *= 1e-9 / np.linalg.norm(w)
    elif kind == "tiny7":
        w = rng.normal(size=3); w *= 3e-8 / np.linalg.norm(w)   # just above sqrt(eps)
    elif kind == "small5":
        w = rng.normal(size=3); w *= 1e-5 / np.linalg.norm(w)
    elif kind == "small3":
        w = rng.normal(size=3); w *= 1e-3 / np.linalg.norm(w)
    elif kind == "big":
        w = rng.normal(size=3); w *= rng.uniform(np.pi, 2 * np.pi - 0.05) / np.linalg.norm(w)
    elif kind == "nearpi":
        w = rng.normal(size=3); w *= (np.pi + rng.uniform(-1e-3, 1e-3)) / np.linalg.norm(w)
    else:
        w = rng.uniform(-np.pi, np.pi, size=3) * rng.uniform(0.05, 1.0)
    t = rng.uniform(-5, 5, size=3)
    f = rng.uniform(0.8, 1.2)
    if kind == "nodist":
        k1 = k2 = 0.0
    else:
        k1, k2 = rng.uniform(-1e-2, 1e-2, size=2)
    # a point in front of the camera (camera looks down -z), |p| up to ~1.2
    z = -rng.uniform(1.0, 10.0)
    if kind == "nearz":
        z = -rng.uniform(1e-3, 1e-2)
    qx, qy = rng.uniform(-1.2, 1.2, size=2) * (-z)
    q = np.array([qx, qy, z])
    R = np.array(rot(w).tolist(), dtype=float)
    X = R.T @ (q - t)
    return np.concatenate([w, t, [f, k1, k2]]), X


def fl(x):
    return float(x)


def main():
    rng = np.random.default_rng(20240)
    kinds = (["generic"] * 120 + ["nodist"] * 30 + ["zero"] * 6 + ["tiny9"] * 12 + ["tiny7"] * 12 +
             ["small5"] * 12 + ["small3"] * 12 + ["big"] * 40 + ["nearpi"] * 16 + ["nearz"] * 10)
    pairs = []
    worst = mp.mpf(0)
    for kind in kinds:
        bal9, X = make_pair(rng, kind)
        q, uv = model(bal9, X)
        Jn = jac_numeric(bal9, X)
        Ja = jac_analytic(bal9, X)
        for i in range(2):
            for j in range(12):
                scale = max(abs(Ja[i][j]), mp.mpf(1))
                worst = max(worst, abs(Jn[i][j] - Ja[i][j]) / scale)
        R = rot(bal9[:3])
        pairs.append({
            "kind": kind,
            "bal9": [fl(x) for x in bal9],
            "X": [fl(x) for x in X],
            "R_rowmajor": [fl(R[i, j]) for i in range(3) for j in range(3)],
            "q": [fl(q[i]) for i in range(3)],
            "uv": [fl(uv[0]), fl(uv[1])],
            "Jc": [fl(Jn[i][j]) for i in range(2) for j in range(9)],
            "Jp": [fl(Jn[i][9 + j]) for i in range(2) for j in range(3)],
        })
    assert worst < mp.mpf(10) ** -25, worst
    print("analytic vs mp.diff worst rel diff:", mp.nstr(worst, 5))

    # a small whole problem with noisy observations, for total_reprojection_error
    n_cam, n_pts = 12, 40
    cams, pts = [], []
    for _ in range(n_cam):
        w = rng.uniform(-0.3, 0.3, size=3)
        cams.append(np.concatenate([w, rng.uniform(-0.5, 0.5, size=3), [rng.uniform(0.9, 1.1)],
                                    rng.uniform(-1e-2, 1e-2, size=2)]))
    for _ in range(n_pts):
        pts.append(np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-9, -4)]))
    row_ptr, pt_idx, uv_obs = [0], [], []
    sums = {1.0: mp.mpf(0), 2.0: mp.mpf(0), 3.0: mp.mpf(0), 1.5: mp.mpf(0)}
    for c in range(n_cam):
        k = int(rng.integers(0, 9)) if c != 5 else 0          # camera 5 sees nothing (ragged)
        idx = rng.choice(n_pts, size=k, replace=False)
        for pi in idx:
            _, uv = model(cams[c], pts[pi])
            obs = [fl(uv[0]) + float(rng.normal(scale=1e-2)), fl(uv[1]) + float(rng.normal(scale=1e-2))]
            pt_idx.append(int(pi)); uv_obs.append(obs)
            for nrm in sums:
                sums[nrm] += abs(uv[0] - mp.mpf(obs[0])) ** mp.mpf(nrm) + abs(uv[1] - mp.mpf(obs[1])) ** mp.mpf(nrm)
        row_ptr.append(len(pt_idx))
    problem = {
        "bal9": [[fl(x) for x in c] for c in cams],
        "pts": [[fl(x) for x in p] for p in pts],
        "row_ptr": row_ptr, "pt_idx": pt_idx, "uv_obs": uv_obs,
        "err_sum": {str(k): fl(v) for k, v in sums.items()},
        "err": {str(k): fl(v ** (1 / mp.mpf(k))) for k, v in sums.items()},
    }
    out = {"meta": {"generator": "tests/golden/gen_golden.py", "mp_dps": mp.mp.dps, "seed": 20240,
                    "model": "q=R(w)X+t; p=-q.xy/q.z; uv=f(1+k1|p|^2+k2|p|^4)p",
                    "jacobian": "mp.diff of the model; Jc row-major 2x9 [w t f k1 k2], Jp row-major 2x3"},
           "pairs": pairs, "problem": problem}
    path = os.path.join(HERE, "snavely_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(pairs), "pairs")


if __name__ == "__main__":
    main()
