#!/bin/bash
# round 5: where the fused verdict kernel's extra time goes (measurement builds: outputs are wrong by construction)
set -u
mkdir -p gpurun_out
N=${FUSED_PAIRS:-1000000}
: > gpurun_out/fused_abl.log
for lib in build/ablate/lib_*.so; do
  echo "== $lib" | tee -a gpurun_out/fused_abl.log
  AQC_LIB=$PWD/$lib timeout 300 python tools/fused_step.py $N 20 2>&1 | grep "round 2\|same bytes" | tee -a gpurun_out/fused_abl.log
done
