#!/bin/bash
# round-3 measurement set: tests, the driver's bench line, CLI end to end (plain / single-member gz / bgzf), kernel profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
: > gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 5000000 2>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 6 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 5000000 --gz --bgzf 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 2000000 --config5 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 2000000 --config5 --gz --gz-level 6 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
python tools/e2e_bench.py --pairs 2000000 --config5 --gz --bgzf 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt
cut -c1-420 gpurun_out/e2e_cli.txt
