#!/bin/bash
# round 5: the fused variant through the golden end-to-end cases, then its rocprofv3 summary (kernel trace + HBM counters, own passes)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fused.py -x -q -k "fused" > gpurun_out/fused_e2e.log 2>&1
echo "tests exit $?" >> gpurun_out/fused_e2e.log
tail -5 gpurun_out/fused_e2e.log
AQC_FUSED=1 EXTRA=--text-step-only bash tools/gpu_profile.sh config3 gpurun_out/profile_fused_text_config3.txt 40
