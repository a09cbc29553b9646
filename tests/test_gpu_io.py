"""BAProblem::write of the RESIDENT problem (c2b_problem_write; src/baproblem.rs:709-785): the .bbal image assembled on
the device -- to_vec of every camera, per-camera counts, big-endian words -- must be byte for byte the file the host
writer (c2b_bal_write, checked against the reference's format in tests/test_host_rows.py) produces from the downloaded
arrays; the text form goes through that host writer."""
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
