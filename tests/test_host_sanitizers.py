"""The host-only C++ of the library (csrc/decimal.hpp, the threaded text IO of csrc/host_baproblem.hpp, the .obj loader,
the batched Poisson darts and the samplers of csrc/host_generate.hpp) under AddressSanitizer + UndefinedBehaviorSanitizer
and, separately, ThreadSanitizer -- the GPU pool has no device sanitizer, these headers need no HIP
(tools/probes/host_asan_harness.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "probes", "host_asan_harness.cpp")
WANT = ["decimal round trips: 0 bad", "text, 1 threads: same", "text, 3 threads: same", "text, 8 threads: same", "binary: same",
        "obj: 3 models, 21600 + 2 + 3 indices", "world points: 20000"]


@pytest.mark.parametrize("flags", ["address,undefined", "thread"])
def test_host_code_is_clean_under_sanitizers(tmp_path, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "harness")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + flags, "-fno-omit-frame-pointer", "-pthread", SRC, "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("this g++ has no " + flags + " runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
                                  TSAN_OPTIONS="halt_on_error=0"))
    text = run.stdout + run.stderr
    assert run.returncode == 0, text[-3000:]
    for line in WANT:
        assert line in text, (line, text[-2000:])
    for bad in ("AddressSanitizer", "runtime error", "ThreadSanitizer", "LeakSanitizer"):
        assert bad not in text, text[-3000:]
