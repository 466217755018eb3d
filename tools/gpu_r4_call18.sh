#!/bin/bash
# round 4, GPU call 18: hybrid .gz -> .gz with SMALL windows (the device's share of a window is consumed right after the pool's)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c18; mkdir -p $O
for CFG in "48 4 12" "64 3 10" "32 4 12" "48 4 8"; do
  set -- $CFG
  AQC_GZ_GROUP=$(($1<<20)) AQC_GZ_WINDOW=$2 AQC_GZ_KEEP=$3 AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench_$1_$2_$3.log 2> $O/bench_$1_$2_$3.err; echo "bench $CFG rc=$?"
  python - $1_$2_$3 <<'PY'
import json, sys
g = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4c18/bench_%s.log" % g).read().strip().splitlines()[-1])
    print("group_window_keep", g, ": file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:420])
except Exception as e:
    print("bench parse failed", e)
PY
  grep -E "device gunzip" $O/bench_$1_$2_$3.err | tail -1
done
