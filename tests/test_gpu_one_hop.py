"""Round 4's device routes against INDEPENDENT restatements, in one hop (they were compared with
their host twins -- product code -- and the twins with numpy / pure-Python restatements: two hops).  Here the device
itself meets the checker:

  * the grid layout on the device          vs  the numpy restatement of src/synthetic.rs:178-258 (tests/_problems.py);
  * the .bal text image written on the device  vs  CPython: every decimal token is Rust's `{}` of its value as
    `Decimal(repr(x))` spells it (tests/test_decimal_text.py: rust_display), and parses back to the same bits;
  * the .bal text parsed on the device     vs  CPython's float() of every token of a file written by Python;
  * the .bbal image written on the device  vs  struct.pack of the words src/baproblem.rs:736-764 writes;
  * cull() on the device                   vs  the pure-Python union-find restatement of src/baproblem.rs:392-550."""
import struct

import numpy as np
import pytest

from _problems import np_grid_layout, random_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0
    return city2ba_amd


@pytest.mark.parametrize("cpb,ppb,blocks,L,inset,ch,ph", [(10, 10, 4, 20.0, 1.0, 1.0, 1.0), (3, 7, 2, 9.5, 0.75, 1.25, 0.5), (7, 3, 11, 13.25, 2.5, 2.0, 3.0)])
def test_device_layout_equals_the_numpy_restatement(c2b, cpb, ppb, blocks, L, inset, ch, ph):
    from city2ba_amd import _lib as Lb
    cams, pts = np_grid_layout(blocks, cpb, ppb, L, inset, ch, ph)
    ba = c2b.BAProblem(0)
    Lb.check(Lb.lib().c2b_problem_synthetic_grid_layout(ba._h, cpb, ppb, blocks, L, inset, ch, ph))
    assert ba._sizes() == (len(cams), len(pts), 0)
    got = ba.cameras()
    # rotation and intrinsics are constants; the translation -dir . pos is arithmetic: every bit
    assert np.array_equal(got.view(np.uint64), cams.view(np.uint64))
    assert np.array_equal(ba.points().view(np.uint64), pts.view(np.uint64))
    ba.close()


def _wild(rng, n):
    w = np.concatenate([rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64),
                        np.ldexp(rng.integers(0, 4096, n // 4).astype(np.float64), -rng.integers(0, 60, n // 4)),
                        10.0 ** rng.integers(-300, 300, n // 8), [0.0, -0.0, 1e22, 1e23, 5e-324, 1.7976931348623157e308, 0.30000000000000004, 1.0, -1.0]])
    w = w[np.isfinite(w)]
    rng.shuffle(w)
    return w


def test_device_text_writer_against_cpython_token_by_token(c2b, tmp_path):
    from test_decimal_text import rust_display
    rng = np.random.default_rng(77)
    P = random_problem(53, 611, 7, seed=5, noise=1e-3, empty_every=6)
    uv, pts, bal9 = P["uv"].copy(), P["pts"].copy(), P["bal9"].copy()
    w = _wild(rng, 6000)
    uv.ravel()[:min(uv.size, 4000)] = w[:min(uv.size, 4000)]
    pts.ravel()[:900] = w[4000:4900]
    bal9[:, 3:].ravel()[:300] = w[5000:5300]                   # not the rotation: the file holds to_vec of the device state
    ba = c2b.BAProblem.from_bal(bal9, pts, P["row_ptr"], P["pt_idx"], uv)
    assert ba.options()["host_text"] == 0
    path = tmp_path / "dev.bal"
    ba.write(str(path))
    bal_dev = ba.cameras_bal()                                  # to_vec of the resident cameras: what the file must hold
    ba.close()
    lines = path.read_text().split("\n")
    n_cam, n_pts, n_obs = len(bal9), len(pts), len(P["pt_idx"])
    assert lines[0] == "%d %d %d" % (n_cam, n_pts, n_obs) and lines[-1] == "" and len(lines) == 2 + n_obs + n_cam + n_pts
    counts = np.diff(P["row_ptr"].astype(np.int64))
    cam_of = np.repeat(np.arange(n_cam), counts)
    for o in range(n_obs):                                      # `cam pt u v` (src/baproblem.rs:718-722)
        assert lines[1 + o] == "%d %d %s %s" % (cam_of[o], P["pt_idx"][o], rust_display(uv[o, 0]), rust_display(uv[o, 1])), o
    for c in range(n_cam):                                      # nine values on ONE line (:724-726)
        assert lines[1 + n_obs + c] == " ".join(rust_display(v) for v in bal_dev[c]), c
    for p in range(n_pts):
        assert lines[1 + n_obs + n_cam + p] == " ".join(rust_display(v) for v in pts[p]), p
    # ... and every token reads back, through CPython, to the bits it came from
    toks = np.array([float(t) for ln in lines[1:1 + n_obs] for t in ln.split(" ")[2:]])
    assert np.array_equal(toks.view(np.uint64), uv.ravel().view(np.uint64))


def test_device_text_parser_against_cpython_float(c2b, tmp_path):
    rng = np.random.default_rng(78)
    P = random_problem(41, 300, 6, seed=9, noise=1e-3, empty_every=5)
    n_obs = len(P["pt_idx"])
    w = _wild(rng, 4000)
    spell = ["%r", "%.17g", "%+.16e", "%.20f"]
    toks_uv = []
    for k in range(2 * n_obs):
        v = float(w[k % len(w)])
        fmt = spell[k % 4] if abs(v) < 1e15 or k % 4 != 3 else "%r"
        t = fmt % v
        if len(t.lstrip("+-").replace(".", "").lstrip("0").split("e")[0]) > 19:      # the device declines > 19 significant digits
            t = "%r" % v
        toks_uv.append(t)
    counts = np.diff(P["row_ptr"].astype(np.int64))
    cam_of = np.repeat(np.arange(len(counts)), counts)
    lines = ["%d %d %d" % (len(P["bal9"]), len(P["pts"]), n_obs)]
    for o in range(n_obs):
        lines.append("%d %d %s %s" % (cam_of[o], P["pt_idx"][o], toks_uv[2 * o], toks_uv[2 * o + 1]))
    lines += [" ".join("%r" % float(v) for v in row) for row in P["bal9"]]
    lines += [" ".join("%.17g" % float(v) for v in row) for row in P["pts"]]
    path = tmp_path / "py.bal"
    path.write_text("\n".join(lines) + "\n")
    c2b.set_default_options(text_device_strict=True, text_device_min_bytes=0)        # the device parser or an error
    ba = c2b.BAProblem.from_file(str(path))
    want_uv = np.array([float(t) for t in toks_uv]).reshape(-1, 2)
    assert np.array_equal(ba.observations().view(np.uint64), want_uv.view(np.uint64))
    assert np.array_equal(ba.cameras_bal().view(np.uint64), P["bal9"].view(np.uint64))
    assert np.array_equal(ba.points().view(np.uint64), P["pts"].view(np.uint64))
    assert np.array_equal(ba.row_ptr, P["row_ptr"]) and np.array_equal(ba.pt_idx, P["pt_idx"])
    ba.close()


def test_device_bbal_image_against_struct_pack(c2b, tmp_path):
    P = random_problem(37, 420, 8, seed=21, noise=1e-2, empty_every=4)
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    path = tmp_path / "dev.bbal"
    ba.write(str(path))
    bal_dev = ba.cameras_bal()
    ba.close()
    n_cam, n_pts, n_obs = len(P["bal9"]), len(P["pts"]), len(P["pt_idx"])
    want = [struct.pack(">QQQ", n_cam, n_pts, n_obs)]          # src/baproblem.rs:736-764: big-endian words
    for c in range(n_cam):
        a, b = int(P["row_ptr"][c]), int(P["row_ptr"][c + 1])
        want.append(struct.pack(">Q", b - a))
        for o in range(a, b):
            want.append(struct.pack(">Qdd", int(P["pt_idx"][o]), P["uv"][o, 0], P["uv"][o, 1]))
    want += [struct.pack(">9d", *row) for row in bal_dev] + [struct.pack(">3d", *row) for row in P["pts"]]
    assert path.read_bytes() == b"".join(want)
    # the same bytes read back on the device
    back = c2b.BAProblem.from_file(str(path))
    assert np.array_equal(back.observations().view(np.uint64), P["uv"].view(np.uint64)) and np.array_equal(back.pt_idx, P["pt_idx"])
    assert np.array_equal(back.cameras_bal().view(np.uint64), bal_dev.view(np.uint64))
    back.close()


@pytest.mark.parametrize("seed,faithful", [(0, True), (1, True), (2, False), (3, True), (4, False)])
def test_device_cull_against_the_python_union_find(c2b, seed, faithful):
    from test_host_rows import _py_cull
    rng = np.random.default_rng(100 + seed)
    n_cam, n_pts = int(rng.integers(30, 80)), int(rng.integers(60, 200))
    # a few clusters plus unseen points (so that the observation-filter quirk of :523 matters)
    rows, tag = [], 0
    for c in range(n_cam):
        k = int(rng.integers(0, 9))
        lo = (c * 3) % max(1, n_pts - 30)
        pts_c = sorted(set(int(x) for x in rng.integers(lo, lo + 30, size=k)))
        rows.append([(p, (tag := tag + 1)) for p in pts_c])
    row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    pt_idx = np.array([p for r in rows for (p, _) in r], dtype=np.uint64)
    uv = np.array([[float(t), -float(t)] for r in rows for (_, t) in r]).reshape(-1, 2)
    cams = rng.normal(size=(n_cam, 15))
    cams[:, :9] = np.eye(3).ravel()
    pts = rng.normal(size=(n_pts, 3))
    kc, kp, new_rows = _py_cull(n_cam, n_pts, [list(r) for r in rows], faithful)
    ba = c2b.BAProblem.from_visibility(cams, pts, row_ptr, pt_idx, uv)
    ba.cull(faithful)
    assert ba.num_cameras() == len(kc) and ba.num_points() == len(kp)
    assert np.array_equal(ba.cameras(), cams[kc]) and np.array_equal(ba.points(), pts[kp])
    want_ptr = np.concatenate([[0], np.cumsum([len(r) for r in new_rows])]).astype(np.uint64)
    assert np.array_equal(ba.row_ptr, want_ptr)
    assert np.array_equal(ba.pt_idx, np.array([p for r in new_rows for (p, _) in r], dtype=np.uint64))
    assert np.array_equal(ba.observations()[:, 0], np.array([float(t) for r in new_rows for (_, t) in r]))
    ba.close()
