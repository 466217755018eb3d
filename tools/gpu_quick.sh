#!/bin/bash
# stream waits sleeping (default) vs spinning: the HBM-resident device step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in block spin block spin; do
  AQC_SYNC=$mode timeout 300 python bench.py --device-only --device-steps 20 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['device_step_mreads_s'], d['device_step']['ms_per_step'], d['roofline']['kernel_ms'])"
done
