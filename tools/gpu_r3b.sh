#!/bin/bash
# the CLI path cold (plain and single-member .gz made with gzip -1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python tools/e2e_bench.py --pairs 5000000 > gpurun_out/e2e_plain.log 2>&1; echo "plain rc=$?"; tail -1 gpurun_out/e2e_plain.log
timeout 500 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 > gpurun_out/e2e_gz1.log 2>&1; echo "gz1 rc=$?"; tail -1 gpurun_out/e2e_gz1.log
