import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library's default arithmetic is fp32 (aas_set_precision(0)).  Tests that do not take a `precision` fixture were written
# against - and deliberately exercise - the split-bf16 fast mode's machinery (operand planes, plane-emitting BPTT, row-major
# weight-gradient GEMM); the suite therefore defaults to that mode, and every parity test that matters runs in BOTH modes through
# its `precision` fixture.  Child processes (multi-rank tests) inherit the variable.
os.environ.setdefault("AAS_PRECISION", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
