#!/bin/bash
# host-only: file write variants on the GPU box's filesystem (tools/ubench/io_probe.cpp); "mmap" = only the write() baseline + shared mappings
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
g++ -O2 -std=c++17 -pthread tools/ubench/io_probe.cpp -o /tmp/io_probe || exit 1
timeout 300 /tmp/io_probe /tmp 2 ${1:-mmap} > gpurun_out/io_probe_mmap.txt 2>&1; echo "rc=$?"; cat gpurun_out/io_probe_mmap.txt
