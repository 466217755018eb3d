#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
GZ_MATRIX='1:2048:4 8:2048:16 16:2048:32 20:2048:40 20:2048:64 32:2048:64' bash tools/gpu_gzrate.sh 1536 | grep -E "inflate" 
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy')}, d['pipe_threads']['seconds'])"; }
timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show
for i in 1 2; do timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --reuse --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show; done
rm -rf /tmp/e2e_gz1
timeout 300 python bench.py --pipe-runs 0 --device-steps 3 --gz-runs 4 --no-pmc --cpu-sample 0 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'gz', d['file_to_file_gz']['mreads_s'], d['file_to_file_gz']['seconds'])"
