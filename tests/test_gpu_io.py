"""BAProblem::write / from_file of the RESIDENT problem (c2b_problem_write, c2b_problem_read; src/baproblem.rs:580-785): both
file forms are assembled / taken apart on the device -- the .bbal image (to_vec of every camera, per-camera counts,
big-endian words) and the .bal decimal text (shortest round-trip digits out, correctly rounded in, observations in any
order) -- and must be byte for byte / bit for bit what the host writer and parser (c2b_bal_write / c2b_bal_read, checked
against the reference's formats in tests/test_host_rows.py) make of the same arrays."""
import numpy as np
import pytest

from _problems import random_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2b():
    import __graft_entry__ as entry
    entry.build()
    import city2ba_amd
    assert city2ba_amd.device_count() > 0
    return city2ba_amd


def _host_file(c2b, ba, path, fmt):
    from city2ba_amd.baproblem import write_bal
    write_bal(path, ba.cameras_bal(), ba.points(), ba.row_ptr, ba.pt_idx, ba.observations(), fmt)
    return open(path, "rb").read()


@pytest.mark.parametrize("empty_every", [0, 3])
def test_resident_write_equals_the_host_writer_byte_for_byte(c2b, tmp_path, empty_every):
    P = random_problem(57, 900, 11, seed=5, noise=1e-3, empty_every=empty_every)
    ba = c2b.BAProblem.from_bal(P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    for fmt, ext in (("binary", "bbal"), ("text", "bal"), (None, "bbal"), (None, "bal")):
        a, b = tmp_path / ("dev." + ext), tmp_path / ("host." + ext)
        ba.write(str(a), fmt)
        assert open(a, "rb").read() == _host_file(c2b, ba, str(b), fmt)
    # cameras perturbed on the device: the file holds to_vec of the CURRENT state (src/baproblem.rs:189-202)
    c2b.noise.add_noise(ba, 0.01, 0.02, 0.03, 0.004, seed=9)
    a, b = tmp_path / "n_dev.bbal", tmp_path / "n_host.bbal"
    ba.write(str(a))
    assert open(a, "rb").read() == _host_file(c2b, ba, str(b), None)
    back = c2b.BAProblem.from_file(str(a))
    assert np.array_equal(back.row_ptr, ba.row_ptr) and np.array_equal(back.pt_idx, ba.pt_idx)
    assert np.array_equal(back.observations(), ba.observations()) and np.array_equal(back.points(), ba.points())
    with pytest.raises(c2b.City2baError, match="extension"):
        ba.write(str(tmp_path / "x.txt"))
    with pytest.raises(c2b.City2baError):
        ba.write(str(tmp_path / "no_such_dir" / "x.bbal"))
    ba.close()


def test_resident_write_without_observations_and_across_chunks(c2b, tmp_path):
    P = random_problem(5, 40, 3, seed=6)
    e = c2b.BAProblem.from_bal(P["bal9"], P["pts"], np.zeros(6, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 2)))
    a, b = tmp_path / "e_dev.bbal", tmp_path / "e_host.bbal"
    e.write(str(a))
    assert open(a, "rb").read() == _host_file(c2b, e, str(b), None)
    e.close()
    # a file of several 16-MiB chunks written by several threads: 1.1 M observations = 27 MB of records
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    a, b = tmp_path / "g_dev.bbal", tmp_path / "g_host.bbal"
    g.write(str(a))
    assert open(a, "rb").read() == _host_file(c2b, g, str(b), None)
    g.close()


def test_resident_read_equals_the_host_reader_plus_upload(c2b, tmp_path):
    """c2b_problem_read (a .bbal decoded on the device; .bal through the host parser): the resident state equals
    read_bal + from_bal's, bit for bit -- cameras (from_vec), points, graph, observations -- for both formats, with empty
    camera lists, and across chunk boundaries; malformed files come back as statuses"""
    from city2ba_amd.baproblem import read_bal, write_bal
    from city2ba_amd import _lib as L
    P = random_problem(61, 700, 9, seed=12, noise=1e-3, empty_every=5)
    for ext in ("bbal", "bal"):
        path = str(tmp_path / ("in." + ext))
        write_bal(path, P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
        a = c2b.BAProblem.from_file(path)
        b = c2b.BAProblem.from_bal(*read_bal(path))
        assert np.array_equal(a.row_ptr, b.row_ptr) and np.array_equal(a.pt_idx, b.pt_idx)
        for x, y in ((a.cameras(), b.cameras()), (a.cameras_bal(), b.cameras_bal()), (a.points(), b.points()), (a.observations(), b.observations())):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
        if ext == "bbal":
            assert np.array_equal(a.cameras_bal(), P["bal9"]) and np.array_equal(a.observations(), P["uv"])
        assert a.total_reprojection_error(2.0) == b.total_reprojection_error(2.0)
        out = str(tmp_path / ("out." + ext))
        a.write(out)
        assert open(out, "rb").read() == open(path, "rb").read()        # read -> write is the identity on the file
        a.close()
        b.close()
    # several 8-MiB chunks (27 MB of records): the count walk crosses chunk boundaries
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    big = str(tmp_path / "g.bbal")
    g.write(big)
    h = c2b.BAProblem.from_file(big)
    assert np.array_equal(h.row_ptr, g.row_ptr) and np.array_equal(h.pt_idx, g.pt_idx)
    assert np.array_equal(h.observations(), g.observations()) and np.array_equal(h.points(), g.points())
    assert np.array_equal(h.cameras_bal(), g.cameras_bal())
    g.close()
    h.close()
    # malformed input: truncated, a count running past the end, a point index out of range, trailing bytes (allowed)
    data = open(str(tmp_path / "in.bbal"), "rb").read()
    ba = c2b.BAProblem(0)
    def status(blob, name):
        q = str(tmp_path / name)
        open(q, "wb").write(blob)
        return L.lib().c2b_problem_read(ba._h, q.encode(), -1)
    assert status(data[:len(data) - 16], "trunc.bbal") == L.ERR_INVALID_ARGUMENT
    assert status(data[:20], "tiny.bbal") == L.ERR_INVALID_ARGUMENT
    huge = bytearray(data); huge[24:32] = (10 ** 12).to_bytes(8, "big")
    assert status(bytes(huge), "count.bbal") == L.ERR_INVALID_ARGUMENT
    first = next(c for c in range(len(P["row_ptr"]) - 1) if P["row_ptr"][c + 1] > P["row_ptr"][c])
    off = 24 + 8 * (first + 1) + 24 * int(P["row_ptr"][first])
    bad = bytearray(data); bad[off:off + 8] = (700).to_bytes(8, "big")      # point 700 of 700
    assert status(bytes(bad), "index.bbal") == L.ERR_INDEX_OUT_OF_RANGE
    assert status(data + b"\\0" * 24, "trailing.bbal") == L.OK             # nom leaves trailing input unread
    assert L.lib().c2b_problem_read(ba._h, str(tmp_path / "missing.bbal").encode(), -1) == L.ERR_INVALID_ARGUMENT
    ba.close()


def test_text_image_from_the_device_equals_the_host_formatter(c2b, tmp_path, monkeypatch):
    """.bal from the resident problem (r04: csrc/text_kernels.hpp -- every decimal of write_text, src/baproblem.rs:709-733,
    formatted on the device by the functions of csrc/decimal.hpp): the same bytes as the host formatter over the
    downloaded arrays, for values of every magnitude -- subnormals, 1e300 (309 characters), exact ties, -0, NaN, inf --
    for a problem without observations, and for a file of several pinned chunks (58 MB)."""
    rng = np.random.default_rng(21)
    P = random_problem(300, 4000, 7, seed=22, noise=1e-3, empty_every=6)
    n_obs = len(P["pt_idx"])
    wild = np.concatenate([rng.integers(0, 2**64, 6000, dtype=np.uint64).view(np.float64),            # any bit pattern
                           np.ldexp(rng.integers(0, 4096, 3000).astype(np.float64), -rng.integers(0, 60, 3000)),
                           rng.integers(0, 2_000_000, 3000) / 1000.0, 10.0 ** rng.integers(-300, 300, 3000),
                           [0.0, -0.0, np.inf, -np.inf, 1e22, 1e23, 5e-324, 1.7976931348623157e308, 0.30000000000000004]])
    uv = P["uv"].copy()
    uv.ravel()[:min(uv.size, len(wild))] = wild[:min(uv.size, len(wild))]
    pts = P["pts"].copy()
    pts.ravel()[:3000] = wild[-3000:]
    bal9 = P["bal9"].copy()
    bal9[:, 3:6].ravel()[:600] = wild[5000:5600]              # translations only: the file holds to_vec of the device state
    bal9[:, 6:9].ravel()[:600] = wild[9000:9600]
    ba = c2b.BAProblem.from_bal(bal9, pts, P["row_ptr"], P["pt_idx"], uv)
    a, b, h = tmp_path / "dev.bal", tmp_path / "host.bal", tmp_path / "route.bal"
    ba.write(str(a))
    dev = open(a, "rb").read()
    assert dev == _host_file(c2b, ba, str(b), None)
    assert dev.count(b"\n") == 1 + n_obs + len(bal9) + len(pts)
    ba.set_options(host_text=True)                            # rounds 1-3's route: download + host formatter
    assert ba.options()["host_text"] == 1
    ba.write(str(h))
    ba.set_options(host_text=False)
    assert open(h, "rb").read() == dev
    ba.close()
    e = c2b.BAProblem.from_bal(P["bal9"][:5], P["pts"][:40], np.zeros(6, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros((0, 2)))
    e.write(str(a))
    assert open(a, "rb").read() == _host_file(c2b, e, str(b), None)
    e.close()
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    g.write(str(a))
    text = open(a, "rb").read()
    assert len(text) > 50_000_000 and text == _host_file(c2b, g, str(b), None)
    back = c2b.BAProblem.from_file(str(a))
    assert np.array_equal(back.observations(), g.observations()) and np.array_equal(back.points(), g.points())
    assert np.array_equal(back.cameras_bal(), g.cameras_bal()) and np.array_equal(back.pt_idx, g.pt_idx)
    g.close()
    back.close()


def _same_state(a, b):
    assert np.array_equal(a.row_ptr, b.row_ptr) and np.array_equal(a.pt_idx, b.pt_idx)
    for x, y in ((a.cameras(), b.cameras()), (a.cameras_bal(), b.cameras_bal()), (a.points(), b.points()), (a.observations(), b.observations())):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))


def test_text_file_parsed_on_the_device_equals_the_host_parser(c2b, tmp_path, monkeypatch):
    """from_file_text (src/baproblem.rs:580-629) on the device (r04: csrc/text_kernels.hpp -- tokens ranked by a scan,
    every number rounded by csrc/decimal.hpp: Clinger's exact case or Eisel-Lemire): the resident state equals the host
    parser's (strtod) bit for bit.  the option text_device_strict makes a file the device declines an error, so these are the
    device's own results; files it must decline (NaN, more than 19 digits, glued numbers,
    an index out of range) reach the host parser and come back as its result or its error."""
    from city2ba_amd.baproblem import read_bal, write_bal
    rng = np.random.default_rng(31)
    c2b.set_default_options(text_device_min_bytes=0)
    P = random_problem(61, 700, 9, seed=12, noise=1e-3, empty_every=5)
    uv = P["uv"].copy()
    wild = np.concatenate([rng.integers(0, 2**64, 4000, dtype=np.uint64).view(np.float64),
                           np.ldexp(rng.integers(0, 4096, 1000).astype(np.float64), -rng.integers(0, 60, 1000)),
                           10.0 ** rng.integers(-300, 300, 500), [0.0, -0.0, 1e22, 1e23, 5e-324, 1.7976931348623157e308, 0.30000000000000004]])
    wild = wild[np.isfinite(wild)]
    uv.ravel()[:min(uv.size, len(wild))] = wild[:min(uv.size, len(wild))]
    path = str(tmp_path / "w.bal")
    write_bal(path, P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], uv)
    c2b.set_default_options(text_device_strict=True)
    a = c2b.BAProblem.from_file(path)                          # the device parsed it (strict)
    c2b.set_default_options(text_device_strict=False)
    b = c2b.BAProblem.from_bal(*read_bal(path))
    _same_state(a, b)
    assert np.array_equal(a.observations().view(np.uint64), uv.view(np.uint64))          # and the file round-trips exactly
    # other spellings of the same numbers: exponents, explicit signs, leading zeros, tabs and blank lines, 17 digits
    lines = open(path).read().split("\n")
    alt = [lines[0].replace(" ", "\t")]
    for i, ln in enumerate(lines[1:]):
        t = ln.split(" ")
        if len(t) == 4:
            t[2], t[3] = "%.17g" % float(t[2]), "%+.16E" % float(t[3])
            alt.append(("  " if i % 3 else "\r\n") + "\t".join(t))
        elif ln:
            alt.append(" ".join(("%.17e" % float(x)) if k % 2 else ("000" + x if x[0].isdigit() else x) for k, x in enumerate(t)))
    path2 = str(tmp_path / "alt.bal")
    open(path2, "w").write("\n".join(alt) + "\n\n  trailing tokens are left unread 1 2 3\n")
    c2b.set_default_options(text_device_strict=True)
    a2 = c2b.BAProblem.from_file(path2)
    _same_state(a2, b)
    # what the device must decline
    declined = {}
    declined["nan"] = "\n".join(lines[:2] + [lines[2].rsplit(" ", 1)[0] + " NaN"] + lines[3:])
    declined["digits"] = "\n".join(lines[:2] + [lines[2].rsplit(" ", 1)[0] + " 0.12345678901234567890123"] + lines[3:])
    declined["glued"] = "\n".join(lines[:2] + [lines[2].rsplit(" ", 1)[0] + "-1.5"] + lines[3:])
    for name, text in declined.items():
        q = str(tmp_path / (name + ".bal"))
        open(q, "w").write(text)
        with pytest.raises(c2b.City2baError, match="declined"):
            c2b.BAProblem.from_file(q)
    c2b.set_default_options(text_device_strict=False)
    for name in ("digits", "glued"):                           # ... and the host parser takes them
        q = str(tmp_path / (name + ".bal"))
        x, y = c2b.BAProblem.from_file(q), c2b.BAProblem.from_bal(*read_bal(q))
        _same_state(x, y)
        x.close(); y.close()
    bad = lines[:]
    bad[1] = "%d %s" % (len(P["bal9"]), bad[1].split(" ", 1)[1])          # camera index out of range
    q = str(tmp_path / "range.bal")
    open(q, "w").write("\n".join(bad))
    with pytest.raises(c2b.City2baError, match="cam_i < cams.len"):
        c2b.BAProblem.from_file(q)
    q = str(tmp_path / "short.bal")
    open(q, "w").write("\n".join(lines[:len(lines) // 2]))
    with pytest.raises(c2b.City2baError):
        c2b.BAProblem.from_file(q)
    assert a.num_observations() == len(uv)                     # a failed read leaves other problems alone
    for x in (a, a2, b):
        x.close()
    # 58 MB of text across several pinned chunks: written by the device, parsed by the device, equal to the host's parse
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(10, 10, 32, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    path = str(tmp_path / "g.bal")
    g.write(path)
    c2b.set_default_options(text_device_strict=True)
    c2b.set_default_options(text_device_min_bytes=-1)
    d = c2b.BAProblem.from_file(path)
    c2b.set_default_options(text_device_strict=False)
    c2b.set_default_options(host_text=True)
    h = c2b.BAProblem.from_file(path)
    _same_state(d, h)
    assert np.array_equal(d.observations(), g.observations()) and np.array_equal(d.cameras_bal(), g.cameras_bal())
    for x in (g, d, h):
        x.close()


def test_both_file_forms_round_trip_at_the_headline_size(c2b, tmp_path):
    """`synthetic --blocks 128` (BASELINE configs[3]: 660 480 cameras before cull, 19.3 M observations) written and read
    back on the device in both forms -- 564 MB of big-endian words, 972 MB of decimal text (89 M shortest round-trip
    decimals out, 89 M correctly rounded parses in) -- returns every array bit for bit; the noised state (cameras
    through to_vec / from_vec, 17-digit observations) too."""
    import os
    from city2ba_amd import synthetic as S
    g = S.synthetic_grid(10, 10, 128, 20.0, 1.0, 1.0, 1.0, 10.0, False)
    assert g.num_observations() > 19_000_000
    c2b.noise.add_noise(g, 0.0, 0.0, 0.0, 1e-3, seed=5)       # observations with all their digits
    c2b.set_default_options(text_device_strict=True)
    try:
        for ext, size in (("bbal", 500_000_000), ("bal", 900_000_000)):
            path = str(tmp_path / ("g128." + ext))
            g.write(path)
            assert os.path.getsize(path) > size
            back = c2b.BAProblem.from_file(path)
            assert np.array_equal(back.row_ptr, g.row_ptr) and np.array_equal(back.pt_idx, g.pt_idx)
            assert np.array_equal(back.observations().view(np.uint64), g.observations().view(np.uint64))
            assert np.array_equal(back.points().view(np.uint64), g.points().view(np.uint64))
            assert np.array_equal(back.cameras_bal().view(np.uint64), g.cameras_bal().view(np.uint64))
            back.close()
            os.remove(path)
    finally:
        c2b.set_default_options(text_device_strict=False)
    g.close()


@pytest.mark.parametrize("n_cam,n_pts,per_cam,order", [(300, 2000, 9, "point"), (70_001, 3000, 4, "random"), (1, 50, 30, "random")])
def test_text_file_in_any_observation_order_is_sorted_on_the_device(c2b, tmp_path, monkeypatch, n_cam, n_pts, per_cam, order):
    """BAProblem::new pushes observations onto their camera's list in FILE order (src/baproblem.rs:347-353).  The Bundle
    Adjustment in the Large datasets list observations point by point, not camera by camera: the device reader then
    sorts them by camera, stably (csrc/text_kernels.hpp: k_sort_*, one to three 8-bit passes for these camera counts)
    -- the same lists, in the same order, as the host parser's sequential push."""
    from city2ba_amd.baproblem import read_bal, write_bal
    rng = np.random.default_rng(41)
    P = random_problem(n_cam, n_pts, per_cam, seed=42, noise=1e-3, empty_every=7 if n_cam > 1 else 0)
    path = str(tmp_path / "sorted.bal")
    write_bal(path, P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    lines = open(path).read().split("\n")
    n_obs = len(P["pt_idx"])
    obs = lines[1:1 + n_obs]
    if order == "point":                                       # stable by point index: the BAL datasets' order
        obs = [obs[i] for i in np.argsort(P["pt_idx"], kind="stable")]
    else:
        obs = [obs[i] for i in rng.permutation(n_obs)]
    mixed = str(tmp_path / "mixed.bal")
    open(mixed, "w").write("\n".join(lines[:1] + obs + lines[1 + n_obs:]))
    c2b.set_default_options(text_device_min_bytes=0)
    c2b.set_default_options(text_device_strict=True)
    d = c2b.BAProblem.from_file(mixed)
    c2b.set_default_options(text_device_strict=False)
    h = c2b.BAProblem.from_bal(*read_bal(mixed))              # the host parser: sequential per-camera push
    _same_state(d, h)
    assert np.array_equal(d.row_ptr, P["row_ptr"])
    if n_cam > 1:
        assert not np.array_equal(d.pt_idx, P["pt_idx"])      # file order within a camera, not the sorted file's
    d.close(); h.close()


def test_text_reader_edge_files_device_and_host_agree(c2b, tmp_path, monkeypatch):
    """small and odd files: no observations, nothing at all, no trailing newline, signed zeros, values that overflow /
    underflow -- the device's result is the host parser's; a short file, a bad header, an index out of range, a float
    where an index belongs -- the device declines and the host parser words the error"""
    c2b.set_default_options(text_device_min_bytes=0)
    nine = " ".join(["-1.5e+0"] * 9)
    good = {"empty": "0 0 0\n", "noobs": "1 2 0\n" + " ".join(["0.5"] * 9) + "\n1 2 3\n4 5 6\n",
            "one": "1 1 1\n0 0 0.25 -0.5\n" + "\n".join(["1e-3"] * 9) + "\n1 2 3",
            "notrail": "1 1 1 0 0 0.25 -0.5 1 2 3 4 5 6 7 8 9 1 2 3",
            "neg": "1 1 1\n0 0 -0 +0.5\n" + nine + "\n-1 -2 -3\n",
            "range": "1 1 1\n0 0 1e400 -1e400\n" + " ".join(["1e-400"] * 9) + "\n1 2 3\n"}
    bad = {"short": ("1 1 1 0 0 0.25 -0.5 1 2 3 4 5 6 7 8 9 1 2", "bad point block"), "badhdr": ("x 1 1\n", "bad header"),
           "idx": ("1 1 1\n0 1 0.5 0.5\n" + nine + "\n1 2 3\n", "p_i < points.len"),
           "float_idx": ("1 1 1\n0.0 0 0.5 0.5\n" + nine + "\n1 2 3\n", "bad observation 0")}

    def state(b):
        return (b.num_cameras(), b.num_points(), b.num_observations(), b.cameras_bal().tobytes(), b.points().tobytes(),
                b.observations().tobytes(), b.pt_idx.tobytes(), b.row_ptr.tobytes())
    for name, text in good.items():
        path = str(tmp_path / (name + ".bal"))
        open(path, "w").write(text)
        c2b.set_default_options(text_device_strict=True)
        d = c2b.BAProblem.from_file(path)
        c2b.set_default_options(text_device_strict=False)
        c2b.set_default_options(host_text=True)
        h = c2b.BAProblem.from_file(path)
        c2b.set_default_options(host_text=False)
        assert state(d) == state(h), name
        d.close(); h.close()
    for name, (text, words) in bad.items():
        path = str(tmp_path / (name + ".bal"))
        open(path, "w").write(text)
        c2b.set_default_options(text_device_strict=True)
        with pytest.raises(c2b.City2baError, match="declined"):
            c2b.BAProblem.from_file(path)
        c2b.set_default_options(text_device_strict=False)
        with pytest.raises(c2b.City2baError, match=words):
            c2b.BAProblem.from_file(path)


def test_problem_options_are_arguments_of_the_abi_not_environment_variables(c2b, tmp_path, monkeypatch):
    """r05: c2b_problem_options -- defaults, get / set round trip, validation, inheritance by new
    problems through set_default_options, and the environment variables of rounds 1-4 no longer reach the library."""
    from city2ba_amd.baproblem import write_bal
    ba = c2b.BAProblem()
    dflt = dict(host_text=0, text_device_strict=0, read_threads=0, io_threads=0, rank_sort_max_row=0, text_device_min_bytes=-1)
    assert ba.options() == dflt
    ba.set_options(host_text=True, read_threads=5, text_device_min_bytes=123)
    assert ba.options() == dict(dflt, host_text=1, read_threads=5, text_device_min_bytes=123)
    with pytest.raises(c2b.City2baError):
        ba.set_options(read_threads=65)
    with pytest.raises(c2b.City2baError):
        ba.set_options(io_threads=-1)
    with pytest.raises(TypeError):
        ba.set_options(no_such_option=1)
    assert ba.options()["read_threads"] == 5                       # a refused set changes nothing
    ba.close()
    c2b.set_default_options(text_device_strict=True, io_threads=2)
    nb = c2b.BAProblem()
    assert nb.options() == dict(dflt, text_device_strict=1, io_threads=2)
    nb.close()
    c2b.reset_default_options()
    # the old variables are just variables now: with C2B_TEXT_DEVICE_STRICT=1 in the environment a file the device parser
    # declines (glued numbers) still goes to the host parser, and a strict problem refuses it whatever the environment says
    P = random_problem(9, 60, 5, seed=3, noise=1e-3)
    path = str(tmp_path / "p.bal")
    write_bal(path, P["bal9"], P["pts"], P["row_ptr"], P["pt_idx"], P["uv"])
    lines = open(path).read().split("\n")
    lines[1] = lines[1].rsplit(" ", 1)[0] + "-1.5"
    open(path, "w").write("\n".join(lines))
    monkeypatch.setenv("C2B_TEXT_DEVICE_STRICT", "1")
    monkeypatch.setenv("C2B_TEXT_DEVICE_MIN_BYTES", "0")
    ok = c2b.BAProblem.from_file(path)
    assert ok.num_observations() == len(P["pt_idx"])
    ok.close()
    c2b.set_default_options(text_device_strict=True, text_device_min_bytes=0)
    with pytest.raises(c2b.City2baError, match="declined"):
        c2b.BAProblem.from_file(path)
