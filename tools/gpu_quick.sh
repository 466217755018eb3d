#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
g++ -O3 -std=c++17 -pthread tools/ubench/translate_rate.cpp afterqc_amd/csrc/aqc_inflate.cpp -lz -o /tmp/translate_rate && /tmp/translate_rate | tee gpurun_out/translate_rate.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_threads or single_end or accum" 2>&1 | tail -3
