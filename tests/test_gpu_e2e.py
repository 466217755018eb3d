"""GPU end-to-end (-m gpu): `after.py`-compatible runs on the HIP engine reproduce the real
reference's outputs byte for byte (the same golden cases test_host_golden.py runs on the oracle), through the serial
chunk loop, the host cross-check path and the whole-input pipe (one and two contexts)."""
import pytest

import cases
from test_host_golden import MODES, PIPE_MODES, check_case, run_case

pytestmark = pytest.mark.gpu

# cases the pipe takes itself (the others — index files, --qc_only — are routed to the serial loop by seqFilter; .bz2 inputs go
# through the pipe since round 5: libbz2 on its own threads)
PIPE_CASES = [c[0] for c in cases.CASES if "-7" not in c[1] and "--qc_only" not in c[1]]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", [c[0] for c in cases.CASES])
def test_e2e_on_gpu(name, mode, tmp_path, e2e, gpu_engine):
    work, stat = run_case(name, tmp_path, gpu_engine, mode)
    check_case(name, work, stat, e2e)


@pytest.mark.parametrize("mode", list(PIPE_MODES))
@pytest.mark.parametrize("name", [c[0] for c in cases.CASES])
def test_e2e_pipe_on_gpu(name, mode, tmp_path, e2e, gpu_engine):
    info = {}
    work, stat = run_case(name, tmp_path, gpu_engine, mode, info)
    check_case(name, work, stat, e2e)
    assert info["used_pipe"] == (name in PIPE_CASES and PIPE_MODES[mode]["use_pipe"]), (name, info)
