#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_e2e.py -m gpu -x -q -k "gz or cfg5 or gigabyte or gunzip or bz2 or g1" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
GZ_MATRIX='1:2048:4 16:2048:32 20:2048:40 20:2048:64' bash tools/gpu_gzrate.sh 1536 | grep -E "inflate|thread-ms"
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy')}, d['pipe_threads']['seconds'])"; }
for i in 1 2; do timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --bgzf 2>/dev/null | tail -1 | show; done
