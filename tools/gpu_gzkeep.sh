#!/bin/bash
# who decodes how much of a .gz input: AQC_GZ_KEEP (fifths of a group the pool must still have in front of it for the device to be
# given another one; default 2) — the GPU is the busier side of a .gz -> .gz run since round 6
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in ${SWEEP:-"AQC_GZ_KEEP=2" "AQC_GZ_KEEP=6" "AQC_GZ_KEEP=10" "AQC_GZ_KEEP=15" "AQC_GZ_KEEP=2" "AQC_GZ_KEEP=6" "AQC_GZ_KEEP=10" "AQC_GZ_KEEP=15"}; do
  env ${v//,/ } timeout 600 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 1 --gz-runs 5 --no-pmc --no-fused-step --inputs 1 --big-copies 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['file_to_file_gz']
print('$v', 'gz->gz', g['mreads_s'], 'median_s', g['median_seconds'], 'host_only', g['host_only_mreads_s'], 'share', g['gunzip_text_share_from_device'])" | tee -a gpurun_out/gzkeep_ab.log
done
