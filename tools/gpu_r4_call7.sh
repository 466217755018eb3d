#!/bin/bash
# round 4, GPU call 7: hybrid gunzip with the bounded device share + low-priority streams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c7; mkdir -p $O
for MODE in default host; do
  [ $MODE = host ] && export AQC_GZ_DEVICE_IN=0
  AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench_$MODE.log 2> $O/bench_$MODE.err; echo "bench $MODE rc=$?"
  python - $MODE <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4c7/bench_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:700])
except Exception as e:
    print("bench parse failed", e)
PY
  grep -E "gunzip|CPU seconds" $O/bench_$MODE.err | tail -4
done
