#!/bin/bash
# round 4, GPU call 27: where the gunzip consumer's time goes inside read() (.gz -> .gz, hybrid and host alone)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c27; mkdir -p $O
for D in 1 0; do
  AQC_PIPE_DEBUG=1 AQC_GZ_DEVICE_IN=$D timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 2 > $O/bench_d$D.log 2> $O/bench_d$D.err; echo "bench device_in=$D rc=$?"
  python - $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4c27/bench_d%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("device_in", sys.argv[1], "file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:420])
PY
  grep -E "gunzip consumer|pipe: gunzip —|CPU seconds|readers / workers done|writers done" $O/bench_d$D.err | tail -12 | cut -c1-330
done
