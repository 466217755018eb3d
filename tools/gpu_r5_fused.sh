#!/bin/bash
# round 5: the fused verdict + placement + copy variant (AQC_FUSED=1): parity tests, then the device text step both ways
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/fused_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/fused_tests.log
tail -15 gpurun_out/fused_tests.log
N=${FUSED_PAIRS:-5000000}
timeout 600 python tools/fused_step.py $N 10 > gpurun_out/fused_step.log 2>&1
echo "step exit $?" >> gpurun_out/fused_step.log
cat gpurun_out/fused_step.log
for lib in build/ablate/lib_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib" | tee -a gpurun_out/fused_step.log
  AQC_LIB=$PWD/$lib timeout 600 python tools/fused_step.py $N 10 2>&1 | grep -v "^round [01]" | tee -a gpurun_out/fused_step.log
done
