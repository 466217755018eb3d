#!/bin/bash
# round 4, GPU call 33: bench's .gz keys with the cold / warm rule (host alone and warm pool + device), the gz tests, the CLI on the same kind of input
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c33; mkdir -p $O
timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c33/bench.log").read().strip().splitlines()[-1])
print("file_to_gz", json.dumps(d.get("file_to_gz"))[:80]); print("file_to_file_gz", json.dumps(d.get("file_to_file_gz")))
PY
timeout 600 python -m pytest tests -m gpu -q -x -k "gz or gzip or gunzip or bgzf" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -2 $O/pytest_gz.log | cut -c1-200
timeout 300 python tools/e2e_bench.py --pairs 2000000 --gz --gz-level 2 2> $O/cli.err | tail -1 | cut -c1-420
