import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suite runs in the LIBRARY'S DEFAULT arithmetic, fp32 (aas_set_precision(0); AAS_PRECISION in the environment moves the
# default for a whole run and is inherited by the child processes of the multi-rank tests).  Parity tests that matter take the
# `precision` fixture (fp32 / split-bf16 fast mode / fp32-equivalent) or `precision2` (fp32 / fast mode: the trainer-level schedule,
# graph, data-parallel and validation tests); tests of the fast mode's own machinery (operand planes, plane-emitting BPTT, row-major
# weight-gradient GEMM) pin mode 1 with `fast_mode`.  Every test leaves the process in the default mode (autouse fixture below).
DEFAULT_PRECISION = int(os.environ.get("AAS_PRECISION", "0"))
PRECISION_IDS = {0: "fp32", 1: "splitbf16", 2: "fp32eq"}


def _ops():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aas_enhancement_amd import ops
    return ops


@pytest.fixture(params=[0, 1, 2], ids=["fp32", "splitbf16", "fp32eq"])
def precision(request):
    _ops().set_precision(request.param)
    return request.param


@pytest.fixture(params=[0, 1], ids=["fp32", "splitbf16"])
def precision2(request, monkeypatch):
    _ops().set_precision(request.param)
    monkeypatch.setenv("AAS_PRECISION", str(request.param))     # child processes (multi-rank tests) run the same mode
    return request.param


@pytest.fixture
def fast_mode():
    _ops().set_precision(1)
    return 1


@pytest.fixture(autouse=True)
def _default_precision_after_every_test():
    yield
    try:
        import torch
        if torch.cuda.is_available():
            from aas_enhancement_amd import ops
            if ops.get_precision() != DEFAULT_PRECISION:
                ops.set_precision(DEFAULT_PRECISION)
    except Exception:  # noqa: BLE001
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A hung rendezvous of a multi-process CPU test must fail, not stall the suite (pytest-timeout, when it is installed)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("gpu") is None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
