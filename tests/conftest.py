import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a report that cannot be built fails the run under test (the CLI only prints a notice: the FASTQ outputs and the statistics
    # are complete by then — round-3 advisory: that notice must not hide a regression from the suites)
    os.environ.setdefault("AQC_REPORT_STRICT", "1")


@pytest.fixture(scope="session")
def fvec():
    """Function-level golden vectors produced by the real reference (tests/golden/make_golden.py)."""
    with gzip.open(os.path.join(GOLDEN, "function_vectors.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def e2e():
    """End-to-end golden records (counters, stat JSON, output digests) of the real reference."""
    with gzip.open(os.path.join(GOLDEN, "e2e_cases.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_engine():
    """One HIP context for the whole session; fails loudly if the library or the GPU is missing."""
    from afterqc_amd import capi
    eng = capi.Engine(0, 3)
    yield eng
    eng.close()
