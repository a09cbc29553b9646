"""The C++ command line builds without a GPU and documents the reference's subcommands and flags
(src/bin/city2ba.rs:33-260); anything that needs the device fails with the library's error, not a crash."""
import subprocess

import pytest


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as entry
    entry.build_hip()
    return entry.build_cli()


def run(cli, *args):
    return subprocess.run([cli] + [str(a) for a in args], capture_output=True, text=True, timeout=120)


def test_help_lists_every_subcommand_and_flag(cli):
    r = run(cli, "--help")
    assert r.returncode == 0
    for sub in ("synthetic", "synthetic-line", "noise", "generate", "ply"):
        assert sub in r.stdout
    flags = {
        "synthetic": ["--blocks", "--cameras-per-block", "--points-per-block", "--max-dist", "--camera-height",
                      "--point-height", "--block-inset", "--block-length"],
        "synthetic-line": ["--cameras", "--points", "--max-dist", "--camera-height", "--point-height", "--point-offset", "--length"],
        "noise": ["--rotation-std", "--translation-std", "--point-std", "--observation-std", "--drift-std", "--drift-strength",
                  "--fixed-drift", "--drift-angle", "--mismatch-chance", "--drop-features", "--split-landmarks",
                  "--join-landmarks", "--sin-strength", "--sin-frequency"],
        "generate": ["--cameras", "--intrinsics-start", "--intrinsics-end", "--points", "--max-dist", "--ground", "--height",
                     "--no-lcc", "--move-to-origin", "--path", "--step-size"],
        "ply": [],
    }
    for sub, fl in flags.items():
        r = run(cli, sub, "--help")
        assert r.returncode == 0, r.stderr
        for f in fl:
            assert f in r.stdout, (sub, f)


def test_argument_errors_do_not_need_a_device(cli, tmp_path):
    r = run(cli)
    assert r.returncode == 1 and "USAGE" in r.stdout
    r = run(cli, "frobnicate")
    assert r.returncode != 0 and "wasn't recognized" in r.stderr
    r = run(cli, "synthetic")
    assert r.returncode != 0 and "required arguments" in r.stderr
    r = run(cli, "synthetic", tmp_path / "a.bal", "--bogus", "1")
    assert r.returncode != 0 and "wasn't expected" in r.stderr
    r = run(cli, "synthetic", tmp_path / "a.bal", "--blocks")
    assert r.returncode != 0 and "requires a value" in r.stderr
    r = run(cli, "synthetic", tmp_path / "a.bal", "--blocks", "x")
    assert r.returncode != 0 and "invalid digit" in r.stderr
    r = run(cli, "generate", tmp_path / "none.obj", tmp_path / "o.bal")
    assert r.returncode != 0 and "Could not open file" in r.stderr
    r = run(cli, "noise", tmp_path / "none.bal", tmp_path / "o.bal")
    assert r.returncode != 0
