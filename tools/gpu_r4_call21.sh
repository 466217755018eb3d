#!/bin/bash
# round 4, GPU call 21: scan with the early exit for over-subscribed codes: exactness + kernel times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipe.py -m gpu -q -x -k "gunzip" > $O/pytest_gz.log 2>&1; echo "pytest gunzip rc=$?"; tail -2 $O/pytest_gz.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -2 $O/gunzip_419_l6_g256.log
timeout 300 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 268435456 > $O/gunzip_1250_l1.log 2>&1; echo "gunzip1250 rc=$?"; tail -2 $O/gunzip_1250_l1.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > /dev/null 2>&1); python tools/pmc_summary.py $O/kt_gunzip 2>/dev/null | grep gzb | cut -c1-160
