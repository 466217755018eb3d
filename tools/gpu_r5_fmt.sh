#!/bin/bash
# round 5: the formatter's tests (text path, spans, fused, irregular records, golden end-to-end cases), then kernel traces of the three steps
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_spans.py tests/test_gpu_fused.py tests/test_gpu_irregular.py tests/test_gpu_e2e.py tests/test_gpu_text_fuzz.py -x -q -m gpu > gpurun_out/fmt_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/fmt_tests.log
tail -4 gpurun_out/fmt_tests.log
KT_FUSED=1 bash tools/gpu_r5_kt.sh
