#!/bin/bash
# host-only: the gzip codec's thread scaling on the GPU box's CPUs, with the per-phase thread times
#   bash tools/gpu_gzrate.sh [MiB]                 the T = 1 .. 128 ladder
#   GZ_MATRIX="32:1024:48 ..." bash tools/gpu_gzrate.sh    threads : section KiB : sections in flight
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
g++ -O3 -std=c++17 -pthread -DAQC_GZ_PROFILE tools/ubench/gz_rate.cpp afterqc_amd/csrc/aqc_inflate.cpp afterqc_amd/csrc/aqc_gunzip.cpp afterqc_amd/csrc/aqc_deflate.cpp -lz -o /tmp/gz_rate 2>/dev/null || exit 1
timeout 600 /tmp/gz_rate ${1:-1536} > gpurun_out/gz_rate.txt 2>&1; echo "rc=$?"; cat gpurun_out/gz_rate.txt
