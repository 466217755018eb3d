#!/bin/bash
# .gz -> .gz through bench.py under a few settings (slots, contexts, chunk size), interleaved twice: the file_to_file_gz line of each
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 0 --no-pmc --no-fused-step --big-copies 0 --inputs 1 --gz-runs 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('file_to_file_gz') or {}
print('%-40s gz %.2f median_s %.4f host_only %s f2gz %s value %.1f  threads %s' % ('$*', g.get('mreads_s',0), g.get('median_seconds',0), g.get('host_only_mreads_s'), (d.get('file_to_gz') or {}).get('mreads_s'), d['value'], {k: v for k, v in (g.get('thread_seconds_last_run') or {}).items()}))"; }
for rep in 1 2; do
  run --slots 3
  run --slots 4
  run --slots 6
  run --contexts 2
  run --chunk-records 65536
  run --chunk-records 262144
done
