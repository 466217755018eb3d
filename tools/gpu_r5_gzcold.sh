#!/bin/bash
# round 5: what a COLD device gunzip costs a one-shot CLI run (.gz in -> .gz out, a fresh process each): host pool alone (the default
# under 4 GiB) against the device-assisted run with its set-up and tear-down timed (AQC_GZ_DEBUG=1)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=${GZ_PAIRS:-3000000}
python tools/e2e_bench.py --pairs $P --gz --gz-level 1 --keep --dir /tmp/aqc_gzcold 2>&1 | tail -1 | tee gpurun_out/gzcold.log
python tools/e2e_bench.py --pairs $P --gz --gz-level 1 --keep --reuse --dir /tmp/aqc_gzcold 2>&1 | tail -1 | tee -a gpurun_out/gzcold.log
for k in 1 2; do
  AQC_GZ_DEVICE_MIN=0 AQC_GZ_DEBUG=1 python tools/e2e_bench.py --pairs $P --gz --gz-level 1 --keep --reuse --dir /tmp/aqc_gzcold 2>&1 | grep "gz dev\|mode" | tee -a gpurun_out/gzcold.log
done
