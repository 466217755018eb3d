#!/bin/bash
# round 4, GPU call 25: device deflate whose emit pass replays the size pass's decisions: gz tests + the .gz bench keys
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c25; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "gz or gzip or compress or bgzf" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -3 $O/pytest_gz.log | cut -c1-300
timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 2 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c25/bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "file_to_gz", json.dumps(d.get("file_to_gz"))[:160], "file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:200])
PY
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 2"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o g -- $B > /dev/null 2>&1); python tools/pmc_summary.py $O/kt 2>/dev/null | grep -E "gz_encode|gz_hist|copyBuffer|fast_filter" | cut -c1-170
