#!/bin/bash
# the -m gpu suite (without the soak) with the tree's library, then the device step of $WORKLOADS with it and with build/ablate/*.so, interleaved
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/pytest-of-* /tmp/aqc_* 2>/dev/null      # (a box may be one of this round's earlier ones: pytest keeps the last three runs' 39 GB each)
df -h /tmp | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --tb=short -k "not soak" > gpurun_out/pytest_ab.log 2>&1; echo "pytest rc=$?"; tail -${TAILN:-40} gpurun_out/pytest_ab.log | cut -c1-300
rm -rf /tmp/pytest-of-* 2>/dev/null; df -h /tmp | tail -1
WORKLOADS="${WORKLOADS:-config5 config2 config3}" REPS=${REPS:-2} bash tools/gpu_quick_ab.sh
