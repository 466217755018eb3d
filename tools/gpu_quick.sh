#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
s=$(date +%s); timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["traffic"], d["roofline"]["traffic_source"]); print(d["file_to_file_gz"])
PY
for i in 1 2; do timeout 400 python tools/e2e_bench.py --pairs 5000000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('wall_s','init_s','pass1_s','pass2_s','stats_s','report_s','close_s')})"; done
