#!/bin/bash
# round 4, GPU call 2: full GPU suite (cooperative verification, one-sync framing, device gunzip v2.1), gunzip rates, bench line with the
# hybrid gunzip, verdict kernel A/B (AQC_COOP_VERIFY 1 vs 0), slots sweep of the pinned -> pinned pipe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -3 $O/gunzip_419_l6_g256.log
timeout 300 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 268435456 > $O/gunzip_1250_l1_g256.log 2>&1; echo "gunzip1250 rc=$?"; tail -3 $O/gunzip_1250_l1_g256.log
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c2/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "roofline", "device_step", "pinned_to_pinned", "file_to_file", "file_to_file_gz", "file_to_gz", "multi_input_file_to_file"):
        print(k, json.dumps(d.get(k))[:700])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "gunzip|CPU seconds" $O/bench.err | tail -12
timeout 600 bash tools/gpu_ablate.sh > $O/ablate_time.log 2>&1; echo "ablate time rc=$?"; cat $O/ablate_time.log
timeout 600 bash tools/gpu_pmc_ablate.sh > $O/ablate_pmc.log 2>&1; echo "ablate pmc rc=$?"; cat $O/ablate_pmc.log
for S in 3 4 6; do
  timeout 300 python bench.py --cpu-sample 0 --no-pmc --gz-runs 0 --inputs 0 --steps 3 --warmup 1 --device-steps 2 --slots $S > $O/slots$S.log 2> $O/slots$S.err
  python - $S <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4c2/slots%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
    print("slots", sys.argv[1], "value", d["value"], "pinned", json.dumps(d["pinned_to_pinned"]), "f2f", json.dumps(d["file_to_file"]["thread_seconds_last_run"]))
except Exception as e:
    print("slots", sys.argv[1], "failed", e)
PY
done
