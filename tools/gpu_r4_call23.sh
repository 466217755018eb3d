#!/bin/bash
# round 4, GPU call 23: .gz -> .gz against the number of chunks in flight (slots, contexts)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c23; mkdir -p $O
for CFG in "--slots 3" "--slots 5" "--contexts 2" "--contexts 2 --slots 4"; do
  T=$(echo $CFG | tr -d ' -')
  timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 $CFG > $O/bench_$T.log 2> $O/bench_$T.err; echo "bench $CFG rc=$?"
  python - $T <<'PY'
import json, sys
g = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4c23/bench_%s.log" % g).read().strip().splitlines()[-1])
    print(g, "value", d["value"], ": file_to_gz", json.dumps(d.get("file_to_gz"))[:120], "file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:330])
except Exception as e:
    print("bench parse failed", e)
PY
done
