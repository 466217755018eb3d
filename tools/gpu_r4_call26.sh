#!/bin/bash
# round 4, GPU call 26: pool front lane strictly first + the gunzip consumer helps with its translation pieces: .gz -> .gz, host alone and hybrid
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c26; mkdir -p $O
for D in 1 0; do
  AQC_GZ_DEVICE_IN=$D timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 2 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench_d$D.log 2> $O/bench_d$D.err; echo "bench device_in=$D rc=$?"
  python - $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4c26/bench_d%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print("device_in", sys.argv[1], "value", d["value"], "file_to_gz", json.dumps(d.get("file_to_gz"))[:60], "file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:520])
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -k "gz or gzip or gunzip or bgzf or gigabyte" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -2 $O/pytest_gz.log | cut -c1-300
