#!/bin/bash
# round 4, GPU call 20: scan takes the Kraft test bits from registers; share test with AQC_GZ_KEEP
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipe.py -m gpu -q -x -k "gunzip or gzip or gigabyte" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -5 $O/pytest_gz.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -3 $O/gunzip_419_l6_g256.log
timeout 300 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 268435456 > $O/gunzip_1250_l1.log 2>&1; echo "gunzip1250 rc=$?"; tail -2 $O/gunzip_1250_l1.log
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c20/bench.log").read().strip().splitlines()[-1])
    print("file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:700])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "device gunzip" $O/bench.err | tail -2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > /dev/null 2>&1); python tools/pmc_summary.py $O/kt_gunzip 2>/dev/null | grep gzb | cut -c1-160
