#!/bin/bash
# the CLI path cold (plain and single-member .gz made with gzip -1), after the pipe / text-driver GPU tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipe.py tests/test_text_driver.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 400 python tools/e2e_bench.py --pairs 5000000 > gpurun_out/e2e_plain.log 2>&1; echo "plain rc=$?"; tail -1 gpurun_out/e2e_plain.log
timeout 500 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 > gpurun_out/e2e_gz1.log 2>&1; echo "gz1 rc=$?"; tail -1 gpurun_out/e2e_gz1.log
timeout 300 python bench.py --cpu-sample 0 > gpurun_out/bench_q.log 2>gpurun_out/bench_q.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_q.log
