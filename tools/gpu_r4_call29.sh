#!/bin/bash
# round 4, GPU call 29: .gz -> .gz, device share policy sweep 2 (group MiB, keep fifths, stream priority)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c29; mkdir -p $O
for CFG in "96 2 1" "96 2 0" "96 0 1" "80 1 1" "112 2 1" "96 3 1"; do
  set -- $CFG
  env AQC_GZ_GROUP=$(($1<<20)) AQC_GZ_KEEP=$2 AQC_GZ_PRIO=$3 AQC_PIPE_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 3 > $O/bench_$1_$2_$3.log 2> $O/bench_$1_$2_$3.err; echo "bench group $1 keep $2 prio $3 rc=$?"
  python - $1_$2_$3 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4c29/bench_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("file_to_file_gz")
print(sys.argv[1], "file_to_file_gz", g["mreads_s"], "share", g["gunzip_text_share_from_device"], "seconds", g["seconds"])
PY
  grep -E "gunzip consumer" $O/bench_$1_$2_$3.err | sed -n 5,6p | cut -c42-200
done
