#!/bin/bash
# round 4, GPU call 16: hybrid .gz -> .gz against the device's group size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c16; mkdir -p $O
for G in 32 64 96; do
  AQC_GZ_GROUP=$((G<<20)) AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench_g$G.log 2> $O/bench_g$G.err; echo "bench g$G rc=$?"
  python - $G <<'PY'
import json, sys
g = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4c16/bench_g%s.log" % g).read().strip().splitlines()[-1])
    print("group", g, "MiB: file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:420])
except Exception as e:
    print("bench parse failed", e)
PY
  grep -E "device gunzip" $O/bench_g$G.err | tail -1
done
