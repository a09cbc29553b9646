"""CPU tests of the host-side rows next to the hot path (C++ behind the C ABI; no GPU needed):
grid/line layout, candidate search, hits_building, cull(), .bal/.bbal IO.  The checkers are independent
pure-Python / numpy restatements of the cited reference lines, on small cases."""
import math
import os
import struct
from decimal import Decimal

import numpy as np
import pytest

import __graft_entry__ as entry
import oracle as O
from _problems import (grid_cameras_points, grid_candidate_pairs, np_candidate_pairs, np_grid_layout,
                       random_problem)

entry.build()
from city2ba_amd import _lib as L  # noqa: E402
from city2ba_amd import synthetic as S  # noqa: E402
import city2ba_amd as c2b_mod  # noqa: E402
from city2ba_amd.baproblem import cull_arrays, read_bal, write_bal  # noqa: E402


# ---- layout (src/synthetic.rs:178-258, 323-344) ---------------------------------------------------------
@pytest.mark.parametrize("kw", [dict(num_blocks=3, cameras_per_block=10, points_per_block=20, block_length=5.0),
                                dict(num_blocks=4), dict(num_blocks=2, cameras_per_block=3, points_per_block=7,
                                                         block_length=11.0, block_inset=2.5, camera_height=1.5,
                                                         point_height=0.5)])
def test_grid_layout_matches_numpy_and_oracle(kw):
    pos, dirs, pts = S.grid_layout(**kw)
    n_cam, n_pts = S.grid_sizes(kw["num_blocks"], kw.get("cameras_per_block", 10), kw.get("points_per_block", 10))
    assert pos.shape == (n_cam, 3) and pts.shape == (n_pts, 3)
    cams_np, pts_np = np_grid_layout(kw["num_blocks"], kw.get("cameras_per_block", 10), kw.get("points_per_block", 10),
                                     kw.get("block_length", 20.0), kw.get("block_inset", 1.0),
                                     kw.get("camera_height", 1.0), kw.get("point_height", 1.0))
    assert np.array_equal(pts, pts_np)
    cams = np.array([O.from_position_direction(pos[i], dirs[i]) for i in range(n_cam)])
    assert np.array_equal(cams, cams_np)


def test_grid_layout_asserts_inset():
    with pytest.raises(L.City2baError) as ei:          # assert!(block_inset * 2. < block_length), :177
        S.grid_layout(2, block_length=2.0, block_inset=1.0)
    assert "Block inset" in str(ei.value)


def test_line_layout():
    pos, dirs, pts = S.line_layout(30, 40, 10.0, 1.0, 1.0, 1.0)
    assert np.array_equal(pos[:, 2], np.arange(30) * 10.0 / 29)
    assert np.all(pos[:, 0] == 0.0) and np.all(pos[:, 1] == 1.0)
    assert np.array_equal(pts[:, 2], (np.arange(40) // 2) * 10.0 / 19)
    assert np.array_equal(pts[:, 0], np.where(np.arange(40) % 2 == 0, -1.0, 1.0))
    assert np.array_equal(dirs[0], O.basis_from_angle_y_deg(180.0))


# ---- candidate search + hits_building (src/synthetic.rs:52-124, 277-280) -----------------------------------
def _py_unique_intersection(p0, p1, q0, q1):
    r = (p1[0] - p0[0], p1[1] - p0[1]); s = (q1[0] - q0[0], q1[1] - q0[1])
    rxs = r[0] * s[1] - r[1] * s[0]
    if rxs == 0.0:
        return None
    qp = (q0[0] - p0[0], q0[1] - p0[1])
    t = qp[0] * (s[1] / rxs) - qp[1] * (s[0] / rxs)
    u = qp[0] * (r[1] / rxs) - qp[1] * (r[0] / rxs)
    if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
        return (p0[0] + t * r[0], p0[1] + t * r[1])
    return None


def _py_hits_building(c, p, L_, inset):
    start, end = (c[0], c[2]), (p[0], p[2])
    bi = lambda x: (int(math.trunc(x[0] / L_)), int(math.trunc(x[1] / L_)))       # noqa: E731
    (cbx, cby), (pbx, pby) = bi(start), bi(end)
    for bx in range(min(cbx, pbx), max(cbx, pbx) + 1):
        for by in range(min(cby, pby), max(cby, pby) + 1):
            ox, oy, be = bx * L_, by * L_, L_ - inset
            sides = [((ox + inset, oy + inset), (ox + inset, oy + be)), ((ox + inset, oy + inset), (ox + be, oy + inset)),
                     ((ox + be, oy + inset), (ox + be, oy + be)), ((ox + inset, oy + be), (ox + be, oy + be))]
            for a, b in sides:
                ip = _py_unique_intersection(start, end, a, b)
                if ip is not None:
                    rad = (end[0] - ip[0]) ** 2.0 + (end[1] - ip[1])       # the reference's un-squared y term
                    if rad >= 0.0 and math.sqrt(rad) > 1e-8:
                        return True
    return False


def test_candidate_pairs_and_occlusion():
    cams, pts = grid_cameras_points(3, cpb=4, ppb=6, L=8.0, inset=1.0)
    ctr = O.centers(cams)
    ci0, pi0 = grid_candidate_pairs(cams, pts, 10.0)
    ci, pi = S.candidate_pairs(ctr, pts, 10.0, n_threads=3)
    assert np.array_equal(ci, ci0) and np.array_equal(pi, pi0)
    ci1, pi1 = np_candidate_pairs(ctr, pts, 10.0)
    assert np.array_equal(ci, ci1.astype(np.uint32)) and np.array_equal(pi, pi1.astype(np.uint32))
    # sub-range
    a, b = S.candidate_pairs(ctr, pts, 10.0, cam_lo=5, cam_hi=17)
    m = (ci >= 5) & (ci < 17)
    assert np.array_equal(a, ci[m]) and np.array_equal(b, pi[m])
    # occlusion filter == python restatement of hits_building
    co, po = S.candidate_pairs(ctr, pts, 10.0, occlusion=True, block_length=8.0, block_inset=1.0, n_threads=2)
    keep = np.array([not _py_hits_building(ctr[c], pts[p], 8.0, 1.0) for c, p in zip(ci, pi)])
    assert 0 < keep.sum() < len(keep)
    assert np.array_equal(co, ci[keep]) and np.array_equal(po, pi[keep])


# ---- cull (src/baproblem.rs:392-550) ---------------------------------------------------------------------------
def _py_cull(n_cam, n_pts, rows, faithful=True):
    """rows: list per camera of (point, tag) ; returns (kept camera ids, kept point ids, rows)"""
    cam_ids, pt_ids = list(range(n_cam)), list(range(n_pts))

    def lcc(rows, nc, np_):
        if nc == 0:
            return list(range(nc)), list(range(np_)), rows
        parent = list(range(nc + np_))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for c, obs in enumerate(rows):
            for (p, _) in obs:
                a, b = find(c), find(nc + p)
                if a != b:
                    parent[b] = a
        sets = [find(i) for i in range(nc + np_)]
        members = {}
        for i, s_ in enumerate(sets):
            members.setdefault(s_, []).append(i)
        best = max(members.values(), key=lambda m: (len(m), -m[0]))
        lid = sets[best[0]]
        kc = [c for c in range(nc) if sets[c] == lid]
        kp = [p for p in range(np_) if sets[nc + p] == lid]
        pmap = {p: i for i, p in enumerate(kp)}
        new_rows = []
        for c in kc:
            flt = (lambda p: sets[p] == lid) if faithful else (lambda p: sets[nc + p] == lid)
            new_rows.append([(pmap[p], t) for (p, t) in rows[c] if flt(p)])
        return kc, kp, new_rows

    def singles(rows, nc, np_):
        kc = [c for c in range(nc) if len(rows[c]) > 3]
        cnt = [0] * np_
        for obs in rows:
            for (p, _) in obs:
                cnt[p] += 1
        kp = [p for p in range(np_) if cnt[p] > 1]
        pmap = {p: i for i, p in enumerate(kp)}
        return kc, kp, [[(pmap[p], t) for (p, t) in rows[c] if p in pmap] for c in kc]

    nc, np_ = n_cam, n_pts
    while True:
        kc, kp, rows = lcc(rows, len(cam_ids), len(pt_ids))
        cam_ids = [cam_ids[c] for c in kc]; pt_ids = [pt_ids[p] for p in kp]
        kc, kp, rows = singles(rows, len(cam_ids), len(pt_ids))
        cam_ids = [cam_ids[c] for c in kc]; pt_ids = [pt_ids[p] for p in kp]
        if len(cam_ids) == nc and len(pt_ids) == np_:
            return cam_ids, pt_ids, rows
        nc, np_ = len(cam_ids), len(pt_ids)


@pytest.mark.parametrize("seed,faithful", [(0, True), (1, True), (2, False), (3, True), (4, False)])
def test_cull_matches_python_restatement(seed, faithful):
    rng = np.random.default_rng(seed)
    n_cam, n_pts = 40, 70
    rows = []
    for c in range(n_cam):
        k = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 8, 12]))
        # two weakly linked clusters + isolated cameras => several components, singletons to peel
        lo, hi = (0, 35) if c < 22 else (30, 70)
        pts_c = rng.choice(np.arange(lo, hi), size=min(k, hi - lo), replace=False)
        rows.append([(int(p), 1000 * c + j) for j, p in enumerate(pts_c)])
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    pt_idx = np.array([p for r in rows for (p, _) in r], dtype=np.uint64)
    tags = np.array([t for r in rows for (_, t) in r], dtype=np.float64)
    uv = np.stack([tags, -tags], axis=1)
    cams = np.arange(n_cam * 15, dtype=np.float64).reshape(n_cam, 15)
    pts = np.arange(n_pts * 3, dtype=np.float64).reshape(n_pts, 3) + 0.5
    c2, p2, rp2, pi2, uv2 = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
    kc, kp, krows = _py_cull(n_cam, n_pts, rows, faithful)
    assert np.array_equal(c2, cams[kc]) and np.array_equal(p2, pts[kp])
    assert list(rp2) == list(np.concatenate([[0], np.cumsum([len(r) for r in krows])]))
    assert list(pi2) == [p for r in krows for (p, _) in r]
    assert list(uv2[:, 0]) == [t for r in krows for (_, t) in r]
    # post-conditions of cull
    if len(c2):
        assert np.all(np.diff(rp2.astype(np.int64)) > 3)
        assert np.all(np.bincount(pi2.astype(np.int64), minlength=len(p2)) > 1)


def test_cull_halves_compose_to_cull():
    """largest_connected_component and remove_singletons as separate calls (the reference exposes both,
    src/baproblem.rs:426, :456): applying them alternately until the sizes stop changing is cull()"""
    rng = np.random.default_rng(11)
    n_cam, n_pts = 60, 90
    rows = [np.sort(rng.choice(np.arange((c % 3) * 30, (c % 3) * 30 + 30), size=int(rng.integers(0, 9)), replace=False)) for c in range(n_cam)]
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    pt_idx = np.concatenate(rows).astype(np.uint64)
    uv = rng.uniform(-1, 1, (len(pt_idx), 2))
    cams = rng.uniform(-1, 1, (n_cam, 9))                       # 9-vector camera rows are opaque payload too
    pts = rng.uniform(-1, 1, (n_pts, 3))
    for faithful in (True, False):
        want = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
        g = (cams, pts, row_ptr, pt_idx, uv)
        while True:
            n = (len(g[0]), len(g[1]))
            g = cull_arrays(*cull_arrays(*g, faithful, step="lcc"), faithful, step="singletons")
            if (len(g[0]), len(g[1])) == n:
                break
        assert all(np.array_equal(a, b) for a, b in zip(g, want))
    one = cull_arrays(cams, pts, row_ptr, pt_idx, uv, step="singletons")
    deg = np.diff(row_ptr.astype(np.int64))
    cnt = np.bincount(pt_idx.astype(np.int64), minlength=n_pts)
    assert len(one[0]) == int((deg > 3).sum()) and len(one[1]) == int((cnt > 1).sum())
    assert np.array_equal(one[0], cams[deg > 3]) and np.array_equal(one[1], pts[cnt > 1])
    with pytest.raises(L.City2baError):
        cull_arrays(cams, pts, row_ptr, pt_idx, uv, step="nope")


def test_cull_property_small_graphs():
    """hypothesis over small graphs (where components, ties and the faithful filter's index aliasing all occur
    often): c2b_cull == the Python restatement, and the result is a fixed point"""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    graphs = st.integers(1, 9).flatmap(lambda nc: st.integers(1, 12).flatmap(lambda npt: st.tuples(
        st.just(nc), st.just(npt),
        st.lists(st.lists(st.integers(0, npt - 1), max_size=7, unique=True), min_size=nc, max_size=nc), st.booleans())))

    @settings(max_examples=300, deadline=None)
    @given(graphs)
    def run(g):
        n_cam, n_pts, obs, faithful = g
        rows = [[(p, 100 * c + j) for j, p in enumerate(r)] for c, r in enumerate(obs)]
        row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
        pt_idx = np.array([p for r in rows for (p, _) in r], dtype=np.uint64)
        tags = np.array([t for r in rows for (_, t) in r], dtype=np.float64)
        uv = np.stack([tags, -tags], axis=1).reshape(-1, 2)
        cams = np.arange(n_cam * 15, dtype=np.float64).reshape(n_cam, 15)
        pts = np.arange(n_pts * 3, dtype=np.float64).reshape(n_pts, 3) + 0.5
        out = cull_arrays(cams, pts, row_ptr, pt_idx, uv, faithful)
        kc, kp, krows = _py_cull(n_cam, n_pts, rows, faithful)
        assert np.array_equal(out[0], cams[kc].reshape(-1, 15)) and np.array_equal(out[1], pts[kp].reshape(-1, 3))
        assert list(out[3]) == [p for r in krows for (p, _) in r]
        assert list(out[4][:, 0]) == [t for r in krows for (_, t) in r]
        again = cull_arrays(*out, faithful)
        assert all(np.array_equal(a, b) for a, b in zip(out, again))

    run()


def test_cull_fixed_point_and_empty():
    P = random_problem(30, 300, 9, seed=4)                 # every point seen once => everything is culled
    out = cull_arrays(P["cams15"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    assert len(out[0]) == 0 and len(out[1]) == 0 and len(out[3]) == 0
    cams, pts = grid_cameras_points(2, cpb=4, ppb=6, L=8.0)
    ci, pi = grid_candidate_pairs(cams, pts, 10.0)
    uv, keep = O.visibility_pairs(cams, pts, ci, pi, 10.0)
    ci, pi, uv = ci[keep == 1], pi[keep == 1], uv[keep == 1]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(ci, minlength=len(cams)))]).astype(np.uint64)
    out = cull_arrays(cams, pts, row_ptr, pi.astype(np.uint64), uv)
    assert 0 < len(out[0]) <= len(cams) and 0 < len(out[3]) <= len(pi)
    again = cull_arrays(*out)
    assert all(np.array_equal(a, b) for a, b in zip(out, again))
    e = cull_arrays(np.zeros((0, 15)), np.zeros((0, 3)), np.zeros(1, dtype=np.uint64), [], np.zeros((0, 2)))
    assert len(e[0]) == 0 and len(e[1]) == 0 and list(e[2]) == [0]


# ---- .bal / .bbal (src/baproblem.rs:580-786) -----------------------------------------------------------------------
def _rust_display(x):
    """Rust `{}` for f64: shortest round-trip digits, fixed notation, no trailing '.0'."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(x)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    if s in ("0", "-0"):
        return "-0" if math.copysign(1.0, x) < 0 else "0"
    return s


def _small_file_problem():
    P = random_problem(6, 40, 5, seed=8, noise=1e-3, empty_every=4)
    bal9 = P["bal9"].copy()
    bal9[0] = [0.30000000000000004, 1e-7, -0.0, 1.0, 123456789.125, -2.5e-10, 1.0, 0.0, 1e22]
    return bal9, P["pts"], P["row_ptr"], P["pt_idx"], P["uv"]


def test_write_text_is_byte_exact(tmp_path):
    bal9, pts, row_ptr, pt_idx, uv = _small_file_problem()
    path = tmp_path / "p.bal"
    write_bal(path, bal9, pts, row_ptr, pt_idx, uv)
    lines = ["%d %d %d" % (len(bal9), len(pts), len(pt_idx))]          # write_text, :709-733
    for c in range(len(bal9)):
        for e in range(int(row_ptr[c]), int(row_ptr[c + 1])):
            lines.append("%d %d %s %s" % (c, pt_idx[e], _rust_display(uv[e, 0]), _rust_display(uv[e, 1])))
    for cam in bal9:
        lines.append(" ".join(_rust_display(v) for v in cam))           # 9 values on ONE line
    for p in pts:
        lines.append(" ".join(_rust_display(v) for v in p))
    assert path.read_text() == "\n".join(lines) + "\n"
    assert "0.30000000000000004 0.0000001 -0 1 123456789.125 -0.00000000025 1 0 10000000000000000000000" in path.read_text()


def test_text_and_binary_roundtrip(tmp_path):
    bal9, pts, row_ptr, pt_idx, uv = _small_file_problem()
    for name in ("p.bal", "p.bbal"):
        path = tmp_path / name
        write_bal(path, bal9, pts, row_ptr, pt_idx, uv)
        b2, p2, r2, i2, u2 = read_bal(path)
        assert np.array_equal(b2, bal9) and np.array_equal(p2, pts)        # shortest round-trip => exact
        assert np.array_equal(r2, row_ptr) and np.array_equal(i2, pt_idx) and np.array_equal(u2, uv)
    raw = (tmp_path / "p.bbal").read_bytes()                                # write_binary, :736-764
    assert struct.unpack(">QQQ", raw[:24]) == (len(bal9), len(pts), len(pt_idx))
    n0 = int(row_ptr[1] - row_ptr[0])
    assert struct.unpack(">Q", raw[24:32])[0] == n0
    assert len(raw) == 24 + 8 * len(bal9) + 24 * len(pt_idx) + 72 * len(bal9) + 24 * len(pts)
    assert struct.unpack(">d", raw[-8:])[0] == pts[-1, 2]


def test_text_reader_accepts_canonical_bal_layout(tmp_path):
    """Canonical BAL files put one camera value per line and observations in any camera order; the reader
    (nom tokens separated by any whitespace, then BAProblem::new) must take both."""
    path = tmp_path / "c.bal"
    cams = [[0.1, 0.2, 0.3, 1, 2, 3, 1.0, 0.0, 0.0], [0, 0, 0, -1, -2, -3, 1.5, 1e-3, -2e-4]]
    text = "2 3 4\n1 2 0.5 0.25\n0 0 -1.0e-1 2\n1 0 3 4\n0 1 5 6\n"
    text += "\n".join(str(v) for cam in cams for v in cam) + "\n"
    text += "1 2 3\n4 5 6\n7 8 9\n"
    path.write_text(text)
    bal9, pts, row_ptr, pt_idx, uv = read_bal(path)
    assert np.array_equal(bal9, np.array(cams, dtype=float))
    assert list(row_ptr) == [0, 2, 4] and list(pt_idx) == [0, 1, 2, 0]     # per-camera push order
    assert np.array_equal(uv, [[-0.1, 2], [5, 6], [0.5, 0.25], [3, 4]])


def test_threaded_text_io_equals_the_sequential_forms(tmp_path, monkeypatch):
    """Above 1 MB the text writer formats its lines on a pool of threads and the reader tokenises in parallel (r04: 7 s /
    12.7 s for the 1.2 GB of `synthetic --blocks 128` on one thread).  Same bytes out, same arrays in, for every thread
    count; observations in any camera order; and a file only nom's grammar accepts -- numbers glued to each other -- takes
    the sequential parser through the fallback and still parses."""
    rng = np.random.default_rng(4)
    n_cam, n_pts = 3000, 9000
    counts = rng.integers(0, 40, size=n_cam)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    n = int(row_ptr[-1])
    pt_idx = rng.integers(0, n_pts, size=n).astype(np.uint64)
    uv = rng.uniform(-1, 1, size=(n, 2))
    uv[7] = [np.inf, -0.0]
    uv[8] = [1e-310, -123456789012345.67]
    bal9, pts = rng.normal(size=(n_cam, 9)), rng.normal(size=(n_pts, 3)) * 1e3
    files = {}
    for t in ("1", "2", "5", "16"):
        c2b_mod.set_host_io_threads(int(t))
        path = tmp_path / ("w%s.bal" % t)
        write_bal(path, bal9, pts, row_ptr, pt_idx, uv)
        files[t] = path.read_bytes()
        got = read_bal(path)
        for a, b in zip(got, (bal9, pts, row_ptr, pt_idx, uv)):
            assert np.array_equal(a, b)
    assert len(files["1"]) > (1 << 20) and files["1"] == files["2"] == files["5"] == files["16"]
    # observations listed in another camera order: BAProblem::new pushes per camera in file order (stable)
    text = files["1"].decode().split("\n")
    obs = text[1:1 + n]
    perm = rng.permutation(n)
    shuffled = tmp_path / "shuffled.bal"
    shuffled.write_text("\n".join([text[0]] + [obs[k] for k in perm] + text[1 + n:]))
    c2b_mod.set_host_io_threads(4)
    par = read_bal(shuffled)
    c2b_mod.set_host_io_threads(1)
    seq = read_bal(shuffled)
    for a, b in zip(par, seq):
        assert np.array_equal(a, b)
    assert np.array_equal(par[2], row_ptr) and not np.array_equal(par[3], pt_idx)
    cam_of = np.repeat(np.arange(n_cam), counts)
    want = np.concatenate([perm[cam_of[perm] == c] for c in range(n_cam)])      # stable bucket by camera
    assert np.array_equal(par[3], pt_idx[want]) and np.array_equal(par[4], uv[want])
    # numbers glued to each other ("0.5-0.25"): one whitespace token, two numbers for nom -> sequential fallback
    k = next(i for i in range(1, n) if " -" in obs[i].split(" ", 2)[2])
    glued = list(text)
    head, rest = glued[1 + k].split(" ", 2)[:2], glued[1 + k].split(" ", 2)[2]
    glued[1 + k] = " ".join(head) + " " + rest.replace(" -", "-", 1)
    gpath = tmp_path / "glued.bal"
    gpath.write_text("\n".join(glued))
    c2b_mod.set_host_io_threads(4)
    got = read_bal(gpath)
    for a, b in zip(got, (bal9, pts, row_ptr, pt_idx, uv)):
        assert np.array_equal(a, b)
    # too few tokens: the sequential parser words the error
    short = tmp_path / "short.bal"
    short.write_text("\n".join(text[:len(text) // 2]))
    with pytest.raises(Exception, match="ParseError|bad"):
        read_bal(short)


def test_hostile_headers_are_status_codes_not_aborts(tmp_path):
    """A corrupt header must not size an allocation (std::length_error / bad_alloc crossing the C ABI would abort the
    host process): counts larger than the file can hold, and counts that overflow u64, are parse errors."""
    import struct as st
    cases = {
        "huge_obs.bal": "1 1 18446744073709551615\n0 0 0.0 0.0\n" + "0 " * 9 + "\n0 0 0\n",
        "huge_cams.bal": "4611686018427387904 1 1\n0 0 0.0 0.0\n" + "0 " * 9 + "\n0 0 0\n",
        "wraps.bal": "1 1 99999999999999999999999999\n0 0 0.0 0.0\n",
        "huge_pts.bal": "1 3074457345618258602 1\n0 0 0.0 0.0\n" + "0 " * 9 + "\n0 0 0\n",
    }
    for name, text in cases.items():
        path = tmp_path / name
        path.write_text(text)
        with pytest.raises(L.City2baError) as ei:
            read_bal(path)
        assert ei.value.status == -1, name
    for nc, np_, no in [(2 ** 63, 1, 1), (1, 2 ** 62, 1), (1, 1, 2 ** 61)]:
        path = tmp_path / "h.bbal"
        path.write_bytes(st.pack(">QQQ", nc, np_, no) + b"\0" * 200)
        with pytest.raises(L.City2baError) as ei:
            read_bal(path)
        assert ei.value.status == -1


def test_file_errors(tmp_path):
    bal9, pts, row_ptr, pt_idx, uv = _small_file_problem()
    with pytest.raises(L.City2baError) as ei:
        write_bal(tmp_path / "p.txt", bal9, pts, row_ptr, pt_idx, uv)
    assert "unknown file extension txt" in str(ei.value)
    with pytest.raises(L.City2baError) as ei:
        write_bal(tmp_path / "noext", bal9, pts, row_ptr, pt_idx, uv)
    assert "does not have an extension" in str(ei.value)
    bad = tmp_path / "bad.bal"
    bad.write_text("1 1 1\n0 5 0.0 0.0\n" + "0 " * 9 + "\n0 0 0\n")
    with pytest.raises(L.City2baError) as ei:          # assert!(p_i < points.len())
        read_bal(bad)
    assert ei.value.status == -2
    trunc = tmp_path / "t.bal"
    trunc.write_text("1 1 1\n0 0 0.0\n")
    with pytest.raises(L.City2baError):
        read_bal(trunc)
    with pytest.raises(L.City2baError):
        read_bal(tmp_path / "missing.bbal")
    # write_text / write_binary / from_file_text / from_file_binary ignore the extension (src/baproblem.rs:580-764)
    for fmt in ("text", "binary"):
        path = tmp_path / ("explicit_" + fmt)
        write_bal(path, bal9, pts, row_ptr, pt_idx, uv, fmt)
        back = read_bal(path, fmt)
        assert all(np.array_equal(a, b) for a, b in zip(back, (bal9, pts, row_ptr, pt_idx, uv)))
        with pytest.raises(L.City2baError) as ei:
            read_bal(path)
        assert "does not have an extension" in str(ei.value)
    assert (tmp_path / "explicit_binary").read_bytes()[:8] == struct.pack(">Q", len(bal9))
    assert (tmp_path / "explicit_text").read_text().split("\n")[0] == "%d %d %d" % (len(bal9), len(pts), len(pt_idx))


def test_poisson_placement_with_a_callers_hierarchy_equals_its_own():
    """c2b_generate_cameras_poisson_bvh (the downward rays of src/generate.rs:240-262 through a hierarchy the caller built)
    places the same cameras as the entry that builds its own; more samples than one thread's share of rays, so the
    gathering in sample order is exercised."""
    import ctypes as C
    from city2ba_amd import _lib as L
    from city2ba_amd.generate import generate_cameras_poisson
    rng = np.random.default_rng(5)
    n = 40
    xs, zs = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32))
    h = rng.uniform(0, 0.3, (n + 1, n + 1)).astype(np.float32)
    tri = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (i, h[i, j], j), (i + 1, h[i + 1, j], j), (i, h[i, j + 1], j + 1), (i + 1, h[i + 1, j + 1], j + 1)
            tri += [a + b + c, b + d + c]
    tri = np.array(tri, dtype=np.float32)
    pos, dirs = generate_cameras_poisson(tri, 6000, 1.7, 100.0, seed=9)
    assert len(pos) > 4096
    bvh = C.c_void_p()
    L.check(L.lib().c2b_bvh_build(tri.ctypes.data_as(C.c_void_p), len(tri), C.byref(bvh)))
    try:
        cap = 4 * 6000
        p2, d2, m = np.empty((cap, 3)), np.empty((cap, 9)), C.c_int64()
        L.check(L.lib().c2b_generate_cameras_poisson_bvh(tri.ctypes.data_as(C.c_void_p), len(tri), bvh, 6000, 1.7, 100.0, 9, cap,
                                                         p2.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), C.byref(m)))
    finally:
        L.lib().c2b_bvh_free(bvh)
    assert m.value == len(pos) and np.array_equal(p2[:m.value], pos) and np.array_equal(d2[:m.value], dirs)
    # minimum distance of the blue-noise set (the nearest-cells-first rejection must reject exactly what the scan did)
    r = np.sqrt(1.1547005383792515 / 12000) * (n - 0)
    xy = pos[:, [0, 2]]
    from scipy.spatial import cKDTree
    d, _ = cKDTree(xy).query(xy, k=2)
    assert d[:, 1].min() >= r * (1 - 1e-6)
