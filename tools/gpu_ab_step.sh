#!/bin/bash
# interleaved A/B of the device step of one workload under an environment switch: tools/gpu_ab_step.sh VAR config5 [pairs] -> VAR=0 / VAR=1 four times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
VAR=$1; W=${2:-config3}; P=${3:-5000000}; [ $W = config5 ] && P=${3:-3000000}
for v in 0 1 0 1; do
  env $VAR=$v python bench.py --workload $W --pairs $P --cpu-sample 0 --device-only --device-steps 10 --no-pmc --no-fused-step --text-step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $VAR=$v device_step', d['device_step']['ms_per_step'])"
done
