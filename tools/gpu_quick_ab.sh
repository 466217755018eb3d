#!/bin/bash
# the device step of the named workloads (default config3) with the tree's library and every build/ablate/*.so, interleaved (no tests): quick A/Bs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/quick_ab.log
for W in ${WORKLOADS:-config3}; do P=5000000; [ $W = config5 ] && P=3000000
  echo "## $W" | tee -a gpurun_out/quick_ab.log
  REPS=${REPS:-2} bash tools/gpu_libs_ab.sh --workload $W --pairs $P | tee -a gpurun_out/quick_ab.log
done
