#!/bin/bash
# round 4, GPU call 3: full GPU suite with the LDS-table gunzip decoder + cheap scan survivors, gunzip rates, bench (hybrid policy fixed),
# verdict kernel A/B, config-4-size soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c3; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -3 $O/gunzip_419_l6_g256.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 33554432 > $O/gunzip_419_l6_g32.log 2>&1; echo "gunzip419 g32 rc=$?"; tail -2 $O/gunzip_419_l6_g32.log
timeout 300 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 268435456 > $O/gunzip_1250_l1_g256.log 2>&1; echo "gunzip1250 rc=$?"; tail -2 $O/gunzip_1250_l1_g256.log
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c3/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "roofline", "device_step", "pinned_to_pinned", "file_to_file", "file_to_file_gz", "file_to_gz", "multi_input_file_to_file"):
        print(k, json.dumps(d.get(k))[:800])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "gunzip|CPU seconds" $O/bench.err | tail -14
timeout 600 bash tools/gpu_ablate.sh > $O/ablate_time.log 2>&1; echo "ablate time rc=$?"; cat $O/ablate_time.log
timeout 600 bash tools/gpu_pmc_ablate.sh > $O/ablate_pmc.log 2>&1; echo "ablate pmc rc=$?"; cat $O/ablate_pmc.log
df -h /tmp | tail -1; free -g | head -2
timeout 900 python tools/soak_config4.py > $O/soak.log 2> $O/soak.err; echo "soak rc=$?"; tail -60 $O/soak.log | head -80; tail -5 $O/soak.err
