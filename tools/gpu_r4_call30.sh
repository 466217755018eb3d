#!/bin/bash
# round 4, GPU call 30: .gz -> .gz with three decoder lanes per file and the new defaults (groups of <= 96 MiB, keep 2/5): default, two variants, gz tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c30; mkdir -p $O
for CFG in "0 -" "80 1" "64 1"; do
  set -- $CFG
  E=""; [ $1 != 0 ] && E="AQC_GZ_GROUP=$(($1<<20)) AQC_GZ_KEEP=$2"
  env $E AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 3 > $O/bench_$1.log 2> $O/bench_$1.err; echo "bench group $1 keep $2 rc=$?"
  python - $1 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4c30/bench_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("file_to_file_gz")
print(sys.argv[1], "file_to_file_gz", g["mreads_s"], "share", g["gunzip_text_share_from_device"], "seconds", g["seconds"])
PY
  grep -E "gunzip consumer" $O/bench_$1.err | sed -n 5,6p | cut -c42-200
done
timeout 900 python -m pytest tests -m gpu -q -x -k "gz or gzip or gunzip or bgzf or gigabyte" > $O/pytest_gz.log 2>&1; echo "pytest gz rc=$?"; tail -2 $O/pytest_gz.log | cut -c1-300
