#!/bin/bash
# who uses the host's cores in a .gz -> .gz run (library built with -DAQC_GZ_PROFILE; AQC_PIPE_DEBUG prints the CPU seconds per thread kind)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export AQC_PIPE_DEBUG=1
timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --dir /tmp/e2e_gz1 > gpurun_out/cpu_gz1.log 2>&1; echo "rc=$?"; grep -E "pipe: CPU|pipe: gunzip|^\{\"mode" gpurun_out/cpu_gz1.log | cut -c1-700
timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --reuse --dir /tmp/e2e_gz1 > gpurun_out/cpu_gz1b.log 2>&1; echo "rc=$?"; grep -E "pipe: CPU|pipe: gunzip|^\{\"mode" gpurun_out/cpu_gz1b.log | cut -c1-700
