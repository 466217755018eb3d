#!/bin/bash
# round 4, GPU call 4: hybrid gunzip with the pool-up / device-down windows + batched match copies; verdict kernel with the one-word polyX
# screen and the conditional walk steps; profile of the config-3 device step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -2 $O/gunzip_419_l6_g256.log
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c4/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "roofline", "device_step", "pinned_to_pinned", "file_to_file", "file_to_file_gz", "file_to_gz", "multi_input_file_to_file"):
        print(k, json.dumps(d.get(k))[:800])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "gunzip|CPU seconds" $O/bench.err | tail -12
AQC_GZ_DEVICE_IN=0 timeout 600 python bench.py --cpu-sample 0 --no-pmc --pipe-runs 0 --steps 1 --warmup 1 --inputs 0 --device-steps 2 > $O/bench_hostgz.log 2> $O/bench_hostgz.err; echo "bench host-gz rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c4/bench_hostgz.log").read().strip().splitlines()[-1])
    print("HOST-ONLY gunzip:", json.dumps(d.get("file_to_file_gz"))[:600])
except Exception as e:
    print("bench parse failed", e)
PY
timeout 900 bash tools/gpu_profile.sh config3 gpurun_out/r4c4/profile_config3.txt 30
