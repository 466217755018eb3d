#!/bin/bash
# round 4, GPU call 1: the new device gunzip (exactness + rate), its tests, the verdict kernel's ablation instruction counts, a bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c1; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 300 python tools/gpu_gunzip_dev.py 64 6 > $O/gunzip_64_l6.log 2>&1; echo "gunzip64 rc=$?"; tail -4 $O/gunzip_64_l6.log
timeout 400 python tools/gpu_gunzip_dev.py 419 6 > $O/gunzip_419_l6.log 2>&1; echo "gunzip419 rc=$?"; tail -4 $O/gunzip_419_l6.log
timeout 400 python tools/gpu_gunzip_dev.py 419 1 default 16 1048576 134217728 > $O/gunzip_419_l1.log 2>&1; echo "gunzip419l1 rc=$?"; tail -4 $O/gunzip_419_l1.log
timeout 900 python -m pytest tests/test_gpu_pipe.py -m gpu -q -x -k "gunzip or gzip or gigabyte" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -5 $O/pytest_gz.log
timeout 900 bash tools/gpu_pmc_ablate.sh > $O/ablate_pmc.log 2>&1; echo "ablate rc=$?"; cat $O/ablate_pmc.log
timeout 600 python bench.py --cpu-sample 0 --no-pmc > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.log; tail -5 $O/bench.err
