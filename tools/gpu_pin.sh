#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/ubench/pin_rate.cpp -o /tmp/pin_rate 2>/dev/null || exit 1
timeout 120 /tmp/pin_rate > gpurun_out/pin_rate.txt 2>&1; echo "rc=$?"; cat gpurun_out/pin_rate.txt
