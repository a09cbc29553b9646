"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/city2ba_hip.h
declares, binds with the declared signatures, and refuses to compute without a device (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c2b():
    entry.build()
    import city2ba_amd
    return city2ba_amd


HEADERS = ("city2ba_hip.h", "city2ba_hip_host.h", "city2ba_hip_experimental.h")     # the stable device boundary, the host-side rows, the rest


def _headers_text():
    return "\n".join(open(os.path.join(ROOT, "include", h)).read() for h in HEADERS)


def _header_symbols():
    text = _headers_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c2b_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(c2b):
    from city2ba_amd import _lib
    names = _header_symbols()
    assert len(names) >= 35
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "library does not export " + n
    assert sorted(_lib.SIGNATURES) == names          # the Python binding covers exactly the header


def test_header_is_plain_c_and_links(c2b, tmp_path):
    """The boundary is a C ABI: the header must compile as strict C99 and a C program must link and run against
    the shared library (host-only entry points; no GPU needed)."""
    import subprocess
    from city2ba_amd import _lib
    src = tmp_path / "use_abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "city2ba_hip.h"
#include "city2ba_hip_host.h"
#include "city2ba_hip_experimental.h"
int main(void) {
    int64_t n_cam = 0, n_pts = 0;
    int rc = c2b_synthetic_grid_sizes(10, 10, 4, &n_cam, &n_pts);
    printf("%s rc=%d cams=%lld pts=%lld ws=%lld\n", c2b_version(), rc, (long long)n_cam, (long long)n_pts,
           (long long)c2b_workspace_bytes(1000));
    rc = c2b_partition_cameras(0, 1, 1, 0);
    printf("bad-args rc=%d msg=%s\n", rc, c2b_last_error());
    return (n_cam == 800 && n_pts == 2400 && rc == C2B_ERR_INVALID_ARGUMENT) ? 0 : 1;
}
''')
    exe = tmp_path / "use_abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L" + libdir, "-lcity2ba_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cams=800 pts=2400" in out.stdout and "bad-args rc=-1" in out.stdout


def test_header_cites_reference_lines():
    text = _headers_text()
    assert len(re.findall(r"src/(baproblem|noise|synthetic|generate)\.rs:\d+", text)) >= 15


def test_no_cpu_fallback(c2b):
    if c2b.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(c2b.City2baError) as ei:
        c2b.BAProblem()
    assert ei.value.status == -5


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "city2ba_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "city2ba_oracle" not in src and "orc_" not in src, f
    assert "oracle" not in _headers_text()


def test_workspace_and_partition_host_helpers(c2b):
    L = c2b.lib()
    assert L.c2b_workspace_bytes(0) > 0
    # the in-kernel ticket fold needs one partial per workgroup (>= 4 tiles of 64 observations each)
    assert L.c2b_workspace_bytes(10_000_000) >= 10_000_000 // 256 * 8
    counts = np.array([5, 0, 0, 7, 1, 1, 30, 2, 2, 0, 12], dtype=np.uint64)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    for parts in (1, 2, 3, 4, 8, 16):
        b = np.zeros(parts + 1, dtype=np.int64)
        rc = L.c2b_partition_cameras(row_ptr.ctypes.data_as(C.c_void_p), len(counts), parts,
                                     b.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert b[0] == 0 and b[-1] == len(counts) and np.all(np.diff(b) >= 0)
        per = [int(row_ptr[b[k + 1]] - row_ptr[b[k]]) for k in range(parts)]
        assert sum(per) == int(row_ptr[-1])
        if parts <= 3:
            assert max(per) <= int(row_ptr[-1]) / parts + counts.max()
    assert L.c2b_partition_cameras(None, 3, 2, None) == -1
    assert b"partition_cameras" in L.c2b_last_error()


def test_jacobian_launch_shape_is_host_arithmetic_with_three_classes(c2b):
    """c2b_jacobian_launch_shape (no GPU): one tile per wave in 1 024-thread workgroups below ~6 M observations; above, by the
    store rate of the output set -- 256 x 1 below 6.3 TB/s, 1 024 x 1 between 6.3 and 6.85, 512 x 2 at 6.85 or more and when
    the rate is unknown (0): the table the A/Bs of round 5 produced (profiles/r05_ab_slow_store.txt, r05p_*)."""
    from city2ba_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.c2b_jacobian_launch_shape.argtypes = [C.c_int64, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def shape(n, rate):
        w, t = C.c_int(0), C.c_int(0)
        assert L.c2b_jacobian_launch_shape(n, rate, C.byref(w), C.byref(t)) == 0
        return w.value, t.value
    big = 19_302_494
    for rate, want in ((0.0, (8, 2)), (-5.0, (8, 2)), (5700.0, (4, 1)), (6299.9, (4, 1)), (6300.0, (16, 1)), (6500.0, (16, 1)), (6849.9, (16, 1)),
                       (6850.0, (8, 2)), (7100.0, (8, 2))):
        assert shape(big, rate) == want, rate
    for n in (0, 64, 1_225_066, 2_412_824, 5_999_999):
        for rate in (0.0, 5700.0, 7100.0):
            assert shape(n, rate) == (16, 1)
    assert shape(6_000_000, 7100.0) == (8, 2)
    L.c2b_jacobian_tiles_per_wave.argtypes = [C.c_int64]
    assert L.c2b_jacobian_tiles_per_wave(big) == 2 and L.c2b_jacobian_tiles_per_wave(2_412_824) == 1
    assert L.c2b_jacobian_launch_shape(big, 7000.0, None, None) == 0            # either output may be NULL


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_product_library_has_no_tuning_hooks_and_no_undeclared_entry_points(c2b):
    """every c2b_* symbol the product library exports is declared in the public header and vice versa;
    there is no tuning library any more (rounds 1-5 built one with -DC2B_TUNE; its results are recorded, its code is gone) and
    no source file carries a tuning island."""
    import re
    import __graft_entry__ as entry
    header = re.sub(r"/\*.*?\*/", "", _headers_text(), flags=re.S)
    declared = set(re.findall(r"\b(c2b_[a-z0-9_]+)\s*\(", header))
    product = {s for s in _exported(entry.build_hip()) if s.startswith("c2b_")}
    assert not [s for s in product if "tune" in s], "tuning hooks in the product library"
    assert product <= declared, sorted(product - declared)
    assert declared <= product, sorted(declared - product)
    assert not hasattr(entry, "build_tune")
    csrc = os.path.join(ROOT, "city2ba_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hpp", ".hip", ".inc")):
            assert "C2B_TUNE" not in open(os.path.join(csrc, f)).read(), f
    for rel in ("city2ba_amd/_lib.py", "city2ba_amd/device.py", "city2ba_amd/baproblem.py", "bench.py"):
        assert "tune" not in open(os.path.join(ROOT, rel)).read(), rel


def test_jacobian_stream_policy_follows_the_working_set(c2b):
    """c2b_jacobian_stream_policy is host arithmetic (no GPU): tables = 256 B per camera + 32 B per point, streams =
    4 + 16 B per observation, against the 256 MiB Infinity Cache.  The synthetic grid has 40 B (B + 1) cameras,
    120 B (B + 1) points; observation counts as generated."""
    from city2ba_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    lib.c2b_jacobian_stream_policy.argtypes = [C.c_int64] * 3
    grid = lambda B: (40 * B * (B + 1), 120 * B * (B + 1))
    for B, n_obs, want in ((4, 21_629, 0), (32, 1_225_066, 0), (64, 4_860_000, 0), (100, 11_800_000, 2), (115, 15_600_000, 2),
                           (128, 19_302_494, 3), (150, 26_500_000, 0), (208, 50_868_906, 0)):
        n_cam, n_pts = grid(B)
        assert lib.c2b_jacobian_stream_policy(n_obs, n_cam, n_pts) == want, B
    assert lib.c2b_jacobian_stream_policy(19_302_494, 660_480, 0) == 0          # unknown point count: cached
    # one rank of eight on the headline problem: its cameras, every point, an eighth of the observations -- all of it fits
    assert lib.c2b_jacobian_stream_policy(19_302_494 // 8, 660_480 // 8, 1_981_440) == 0


def test_header_index_lists_every_entry_point_once_and_the_library_reads_no_environment_switch():
    """VERDICT r04 items 6, 7.  The header's top comment carries an index of the entry points by level, generated by
    tools/abi_index.py: the block in the header is exactly what the tool prints, and it names every declared function
    once.  And the library's sources call getenv in ONE place only -- the loader setting C2B_RCCL_LIB: behaviour switches
    are c2b_problem_options."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_index
    header = open(os.path.join(ROOT, "include", "city2ba_hip.h")).read()
    block = abi_index.index_block()
    assert block in header, "run `python tools/abi_index.py --write`"
    body = _headers_text().replace(block, "")
    # the stable device boundary holds at most 120 entry points
    main_only = re.findall(r"(?m)^\s*(?:const\s+)?\w+\s*\**\s*(c2b_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", header.replace(block, ""), flags=re.S))
    assert len(main_only) <= 120, len(main_only)
    declared = re.findall(r"(?m)^\s*(?:const\s+)?\w+\s*\**\s*(c2b_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    listed = re.findall(r"\b([a-z][a-z0-9_]+)\b(?=,|\n| \*/|$)", "\n".join(ln[7:] for ln in block.split("\n") if ln.startswith(" *     ")))
    assert sorted("c2b_" + n for n in listed) == sorted(declared)
    assert len(set(declared)) == len(declared)
    csrc = os.path.join(ROOT, "city2ba_amd", "csrc")
    uses = []
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hpp", ".hip")):
            for k, ln in enumerate(open(os.path.join(csrc, name)).read().split("\n")):
                if "getenv" in ln and not ln.lstrip().startswith("//"):
                    uses.append((name, k + 1, ln.strip()))
    assert len(uses) == 1 and uses[0][0] == "comm_rccl.hpp" and "C2B_RCCL_LIB" in uses[0][2], uses
