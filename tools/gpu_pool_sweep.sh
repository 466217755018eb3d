#!/bin/bash
# pool size sweep on single-member gzip -1 input (made once) and on plain input
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy')}, d['pipe_threads']['seconds'])"; }
timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show
for T in 16 20 24 32 48; do
  echo "AQC_IO_THREADS=$T gz1"; AQC_IO_THREADS=$T timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 --keep --reuse --dir /tmp/e2e_gz1 2>/dev/null | tail -1 | show
done
rm -rf /tmp/e2e_gz1
timeout 600 python tools/e2e_bench.py --pairs 5000000 --keep --dir /tmp/e2e_p 2>/dev/null | tail -1 | show
for T in 16 24 32; do
  echo "AQC_IO_THREADS=$T plain"; AQC_IO_THREADS=$T timeout 600 python tools/e2e_bench.py --pairs 5000000 --keep --reuse --dir /tmp/e2e_p 2>/dev/null | tail -1 | show
done
