#!/bin/bash
# round 4, call 31 (host only): aggregate write rate of 1 .. 16 files at once on the GPU box's filesystem
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
g++ -O2 -std=c++17 -pthread tools/ubench/multi_file_write.cpp -o /tmp/mfw || exit 1
timeout 240 /tmp/mfw /tmp 1.0 > gpurun_out/r4_multi_file_write.txt 2>&1; echo "rc=$?"; cat gpurun_out/r4_multi_file_write.txt
