#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipe.py tests/test_text_driver.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy')}, d['pipe_threads']['seconds'])"; }
timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show
for i in 1 2; do timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --reuse --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show; done
rm -rf /tmp/e2e_gz1
for i in 1 2; do timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --bgzf 2>/dev/null | tail -1 | show; done
timeout 300 python bench.py --pipe-runs 0 --device-steps 3 --gz-runs 3 --no-pmc --cpu-sample 0 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'gz', d['file_to_file_gz']['mreads_s'], d['file_to_file_gz']['seconds'])"
