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
def _library_options_are_back_to_their_defaults_after_every_test():
    """tests choose routes through the ABI's options (city2ba_amd.set_default_options / set_host_io_threads); whatever one
    changed does not leak into the next"""
    yield
    mod = sys.modules.get("city2ba_amd.baproblem")
    if mod is not None:
        mod.reset_default_options()
        if getattr(sys.modules.get("city2ba_amd._lib"), "_lib", None) is not None:
            mod.set_host_io_threads(0)


@pytest.fixture(autouse=True)
def _fold_counters_are_clean_after_every_gpu_test(request):
    """Every in-kernel fold resets the arrival counters of the workspace it ran in (kernels.hpp: ticket_fold); a launch
    that did not would make the next launch on that workspace lose its sum.  Checked after every GPU test, over every
    workspace device.workspace() handed out that is still alive, so that the offender is named."""
    yield
    if request.node.get_closest_marker("gpu") is None or os.environ.get("C2B_NO_SELFCHECK"):
        return
    import torch
    from city2ba_amd import device as D
    torch.cuda.synchronize()
    for ws in D.live_workspaces():
        n = D.workspace_selfcheck(ws)
        assert n == 0, "%d arrival counters left non-zero in a workspace by this test (-1: never initialised)" % n
