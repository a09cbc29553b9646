import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "snavely_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(autouse=True)
def _fold_counters_are_clean_after_every_gpu_test(request):
    """Every in-kernel fold resets its arrival counters (kernels.hpp: ticket_fold); a launch that did not would make
    the 128th launch after it lose its sum.  Checked after every GPU test so that the offender is named."""
    yield
    if request.node.get_closest_marker("gpu") is None or os.environ.get("C2B_NO_SELFCHECK"):
        return
    import ctypes as C
    from city2ba_amd import _lib as L
    n = C.c_int64(-1)
    assert L.lib().c2b_selfcheck_tickets(C.byref(n)) == L.OK
    assert n.value == 0, "%d arrival counters left non-zero by this test" % n.value
