#!/bin/bash
# A/B: stream waits sleeping (default) or spinning, cold CLI on plain and single-member .gz input + the file -> file bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in block spin block spin; do
  export AQC_SYNC=$mode
  timeout 400 python tools/e2e_bench.py --pairs 5000000 --keep --dir /tmp/e2e_p > gpurun_out/e2e_plain_$mode.log 2>&1; echo "$mode plain rc=$?"; tail -1 gpurun_out/e2e_plain_$mode.log | cut -c1-330
  timeout 500 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 1 > gpurun_out/e2e_gz1_$mode.log 2>&1; echo "$mode gz1 rc=$?"; tail -1 gpurun_out/e2e_gz1_$mode.log | cut -c1-330
done
for mode in block spin; do
  export AQC_SYNC=$mode
  timeout 300 python bench.py --cpu-sample 0 --device-steps 3 > gpurun_out/bench_$mode.log 2>gpurun_out/bench_$mode.err; echo "$mode bench rc=$?"; python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_$mode.log').read().strip().splitlines()[-1]); print(d['value'], d['file_to_file'], d['pinned_to_pinned_mreads_s'])"
done
