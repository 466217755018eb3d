#!/bin/bash
# the device step of config 3 with the tree's library and every build/ablate/*.so (tools/build_ablate.sh; or those named in $LIBS), interleaved twice
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
for f in afterqc_amd/csrc/libafterqc_hip.so ${LIBS:-build/ablate/*.so}; do
  AQC_LIB=$PWD/$f python bench.py --cpu-sample 0 --device-only --device-steps 10 --no-pmc --no-fused-step --text-step-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s device_step %.4f  kernel_ms %.4f' % ('$f'[-44:], d['device_step']['ms_per_step'], d['roofline']['kernel_ms']))"
done
done
