#!/bin/bash
# round 5: A/B of library builds (build/ablate/lib_*.so, tools/build_ablate.sh) on the device text step, same box, interleaved twice
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/ab_libs.log
N=${FUSED_PAIRS:-5000000}
for rep in 1 2; do
  for lib in afterqc_amd/csrc/libafterqc_hip.so build/ablate/lib_*.so; do
    echo "== $lib (pass $rep)" | tee -a gpurun_out/ab_libs.log
    AQC_LIB=$PWD/$lib timeout 600 python tools/fused_step.py $N 10 2>&1 | grep "best\|same bytes" | tee -a gpurun_out/ab_libs.log
  done
done
