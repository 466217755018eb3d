#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 400 python tools/e2e_bench.py --pairs 5000000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','init_s','pass1_s','pass2_s','stats_s','report_s','close_s')})"; done
