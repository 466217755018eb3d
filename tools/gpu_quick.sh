#!/bin/bash
# chunk size sweep of the file -> file metric (and .gz -> .gz)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for K in 131072 32768 65536 262144 131072; do
  timeout 300 python bench.py --pipe-runs 0 --device-steps 3 --gz-runs 2 --no-pmc --cpu-sample 0 --steps 6 --warmup 2 --chunk-records $K 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($K, d['value'], d['file_to_file']['seconds_min'], d['file_to_file']['seconds_mean'], 'gz', d['file_to_file_gz']['mreads_s'])"
done
