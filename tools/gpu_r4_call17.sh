#!/bin/bash
# round 4, GPU call 17: device gunzip alone against the group size; hybrid .gz -> .gz against the decoder streams' priority
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c17; mkdir -p $O
for G in 32 64 128; do
  timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 $((G<<20)) > $O/gunzip_419_g$G.log 2>&1; echo "gunzip419 g$G rc=$?"; tail -1 $O/gunzip_419_g$G.log
done
timeout 300 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 $((64<<20)) > $O/gunzip_1250_g64.log 2>&1; echo "gunzip1250 g64 rc=$?"; tail -1 $O/gunzip_1250_g64.log
for P in 1 0; do
  AQC_GZ_PRIO=$P AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 --gz-runs 3 > $O/bench_p$P.log 2> $O/bench_p$P.err; echo "bench prio $P rc=$?"
  python - $P <<'PY'
import json, sys
g = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4c17/bench_p%s.log" % g).read().strip().splitlines()[-1])
    print("prio", g, ": file_to_file_gz", json.dumps(d.get("file_to_file_gz"))[:420])
except Exception as e:
    print("bench parse failed", e)
PY
  grep -E "device gunzip" $O/bench_p$P.err | tail -1
done
