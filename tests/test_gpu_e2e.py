"""GPU end-to-end (-m gpu): `after.py`-compatible runs on the HIP engine reproduce the real
reference's outputs byte for byte (the same 36 golden cases test_host_golden.py runs on the oracle)."""
import pytest

import cases
from test_host_golden import MODES, check_case, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", [c[0] for c in cases.CASES])
def test_e2e_on_gpu(name, mode, tmp_path, e2e, gpu_engine):
    work, stat = run_case(name, tmp_path, gpu_engine, mode)
    check_case(name, work, stat, e2e)
