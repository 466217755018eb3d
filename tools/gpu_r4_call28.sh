#!/bin/bash
# round 4, GPU call 28: .gz -> .gz with the device allowed a larger share (AQC_GZ_KEEP: fifths of a group the pool must still have in front of it)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c28; mkdir -p $O
for CFG in "0 0" "64 0" "64 4" "0 4" "96 2"; do
  set -- $CFG
  G=""; [ $1 != 0 ] && G="AQC_GZ_GROUP=$(($1<<20))"
  env $G AQC_GZ_KEEP=$2 AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 3 > $O/bench_$1_$2.log 2> $O/bench_$1_$2.err; echo "bench group $1 keep $2 rc=$?"
  python - $1_$2 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4c28/bench_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("file_to_file_gz")
print(sys.argv[1], "file_to_file_gz", g["mreads_s"], "share", g["gunzip_text_share_from_device"], "seconds", g["seconds"])
PY
  grep -E "gunzip consumer" $O/bench_$1_$2.err | sed -n 5,6p | cut -c30-260
  grep -E "CPU seconds" $O/bench_$1_$2.err | sed -n 4p | cut -c1-120
done
