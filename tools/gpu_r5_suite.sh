#!/bin/bash
# round 5: the whole -m gpu suite, the device text step both ways (tools/fused_step.py), kernel-trace summaries of the three steps
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/suite.log 2>&1
echo "tests exit $?" >> gpurun_out/suite.log
tail -6 gpurun_out/suite.log
timeout 600 python tools/fused_step.py 5000000 10 > gpurun_out/fused_step.log 2>&1
cat gpurun_out/fused_step.log
KT_FUSED=1 bash tools/gpu_r5_kt.sh
