#!/bin/bash
# the file writers' fallocate-ahead: the ubench, then the pipe with AQC_FALLOC = 0 / 1 / 2 interleaved on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/dma_write_rate tools/ubench/dma_write_rate.hip -lpthread 2>/dev/null && /tmp/dma_write_rate /tmp 1.7 | grep -i "falloc\|whole chunks\|NUMA" | tee gpurun_out/dma_write_rate2.txt
for v in "AQC_FALLOC=0" "AQC_FALLOC=1" "AQC_FALLOC=2" "AQC_FALLOC=0" "AQC_FALLOC=1" "AQC_FALLOC=2"; do
  env $v timeout 600 python bench.py --steps 12 --warmup 2 --cpu-sample 0 --pipe-runs 0 --device-steps 2 --gz-runs 2 --no-pmc --no-fused-step --big-copies ${BIG:-0} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value', d['value'], 'median', d['value_median'], 'best', d['value_best'], 'multi', d['multi_input_file_to_file_mreads_s'], '100M', (d.get('file_to_file_100M') or {}).get('mreads_s'), 'f2gz', d['file_to_gz']['mreads_s'], 'gz', d['file_to_file_gz']['mreads_s'], 'threads', d['file_to_file']['thread_seconds_last_run'])" | tee -a gpurun_out/falloc_ab.log
done
