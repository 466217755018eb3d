#!/bin/bash
# gzip paths on the GPU box: codec micro-benchmark on the host, then the CLI path .gz -> .gz (config 3 and config-5 flavour)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
g++ -O3 -std=c++17 -pthread -Wno-stringop-overflow tests/native/gz_selftest.cpp afterqc_amd/csrc/aqc_inflate.cpp afterqc_amd/csrc/aqc_gunzip.cpp afterqc_amd/csrc/aqc_deflate.cpp -lz -o /tmp/gz_selftest 2>/dev/null
/tmp/gz_selftest bench 2>&1 | tail -14 > gpurun_out/gz_codec_bench.txt; cat gpurun_out/gz_codec_bench.txt
: > gpurun_out/e2e_gz.txt
python tools/e2e_bench.py --pairs 5000000 --gz 2>gpurun_out/e2e_gz.err | tail -1 >> gpurun_out/e2e_gz.txt
python tools/e2e_bench.py --pairs 2000000 --gz --config5 2>>gpurun_out/e2e_gz.err | tail -1 >> gpurun_out/e2e_gz.txt
python tools/e2e_bench.py --pairs 5000000 2>>gpurun_out/e2e_gz.err | tail -1 >> gpurun_out/e2e_gz.txt
python tools/e2e_bench.py --pairs 2000000 --config5 2>>gpurun_out/e2e_gz.err | tail -1 >> gpurun_out/e2e_gz.txt
cut -c1-900 gpurun_out/e2e_gz.txt; tail -5 gpurun_out/e2e_gz.err
