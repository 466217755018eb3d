#!/bin/bash
# the CLI path cold (plain and single-member .gz made with gzip -1), after the end-to-end / text-driver / report GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipe.py tests/test_text_driver.py tests/test_report.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy')}, d['pipe_threads']['seconds'])"; }
for i in 1 2; do timeout 400 python tools/e2e_bench.py --pairs 5000000 2>/dev/null | tail -1 | show; done
timeout 500 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --dir /tmp/e2e_g 2>/dev/null | tail -1 | show
timeout 500 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --reuse --dir /tmp/e2e_g 2>/dev/null | tail -1 | show
